// ds_read_b64_tr_b16 semantics probe (diagnostic): lds[i] = i (16-bit), lane l supplies the address of elements 4l .. 4l+3;
// prints what every lane receives.   hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o /tmp/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ __attribute__((aligned(16))) short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (short)i;
    __syncthreads();
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(lds + (threadIdx.x & 63) * 4));
    *reinterpret_cast<v4s*>(out + threadIdx.x * 4) = r;
}
int main() {
    short* d; short h[256];
    (void)hipMalloc(&d, 512);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    (void)hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
    return 0;
}
