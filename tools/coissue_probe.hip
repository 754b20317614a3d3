// Do the matrix pipe and the VALU of one SIMD overlap ACROSS waves?  (diagnostic, not part of the library)
// One 512-thread workgroup per CU = 2 waves per SIMD. Role by wave: waves 0-3 run an MFMA-only loop (16 independent
// v_mfma_f32_32x32x16_bf16 accumulators, registers only), waves 4-7 a VALU-only loop (independent chains of
// v_exp_f32 + v_fma_f32 / v_max3 / v_cvt_pk, the softmax's instruction mix). Each role is timed alone and together:
//   together == max(alone)  -> the two pipes overlap across waves (a warp-specialised attention could hide its softmax)
//   together == sum(alone)  -> they serialise (SIMD time = matrix time + VALU issue time, as the attention timeline says)
//   hipcc --offload-arch=gfx950 -O3 tools/coissue_probe.hip -o /tmp/coissue_probe && /tmp/coissue_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

// MODE bit 0: MFMA waves active, bit 1: VALU waves active.  VK: 0 = exp2 + add, 1 = fma only, 2 = max3 + cvt_pk mix
template <int MODE, int VK>
__global__ __launch_bounds__(512) void coissue(float* out, int iters, unsigned long long* ticks) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float sink = 0.f;
    if (wave < 4) {
        if (MODE & 1) {
            f32x16 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
            bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) sink += acc[i][0] + acc[i][7];
        }
    } else if (MODE & 2) {
        float x[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) x[k] = (float)(threadIdx.x + k) * 1e-3f;
        float s = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int rep = 0; rep < 2; ++rep) {
                if (VK == 0) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { x[k] = __builtin_amdgcn_exp2f(x[k]) - 1.0f; }
                } else if (VK == 1) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) { x[k] = __builtin_fmaf(x[k], 0.999f, 1e-3f); x[k] = __builtin_fmaf(x[k], 1.001f, -1e-3f); }
                } else {
#pragma unroll
                    for (int k = 0; k < 16; k += 2) {
                        x[k] = __builtin_fmaxf(__builtin_fmaxf(x[k], x[k + 1]), s);
                        x[k + 1] = x[k + 1] + x[k];
                    }
                }
                asm volatile("" : "+v"(x[0]), "+v"(x[5]), "+v"(x[11]));
            }
        }
#pragma unroll
        for (int k = 0; k < 16; ++k) s += x[k];
        sink += s;
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ticks[wave] = t1 - t0;
    if (sink == 1.2345e30f) out[threadIdx.x] = sink;
}

template <int MODE, int VK>
static void run(const char* name, int cus, float* d, unsigned long long* ticks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((coissue<MODE, VK>), dim3(cus), dim3(512), 0, 0, d, 100, ticks);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((coissue<MODE, VK>), dim3(cus), dim3(512), 0, 0, d, iters, ticks);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8];
    (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = (MODE & 1) ? 4.0 * cus * (double)iters * 16 * 32768.0 / (ms * 1e-3) / 1e12 : 0.0;
    // per iteration: MFMA wave 16 MFMAs x 32 pipe cycles = 512; VALU wave 32 (VK 0: +32 sub) / 64 / 32 instructions
    printf("%-44s %8.3f ms  %7.1f TFLOP/s   cycles/iter: mfma wave %6.1f  valu wave %6.1f\n", name, ms, mf,
           (double)h[0] / iters, (double)h[4] / iters);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d;
    unsigned long long* ticks;
    (void)hipMalloc(&d, 4096);
    (void)hipMalloc(&ticks, 64);
    run<1, 0>("MFMA waves alone", cus, d, ticks);
    run<2, 0>("VALU waves alone: 32 x (exp2 + sub)", cus, d, ticks);
    run<3, 0>("both: MFMA + (exp2 + sub)", cus, d, ticks);
    run<2, 1>("VALU waves alone: 64 x fma", cus, d, ticks);
    run<3, 1>("both: MFMA + fma", cus, d, ticks);
    run<2, 2>("VALU waves alone: 16 x (max, max, add)", cus, d, ticks);
    run<3, 2>("both: MFMA + max/add", cus, d, ticks);
    return 0;
}
