// Sustained MFMA throughput probe (diagnostic, not part of the library): every wave issues independent
// v_mfma_f32_16x16x32_bf16 back to back from registers only — the ceiling any GEMM main loop on this part sits under at
// the clocks the chip actually holds.   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int NACC>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e30f) out[threadIdx.x] = s;
}

template <int NACC>
static void run(const char* name, int blocks_per_cu, int cus, float* d) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mfma_loop<NACC><<<cus * blocks_per_cu, 256>>>(d, 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<NACC><<<cus * blocks_per_cu, 256>>>(d, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * 16 * 16 * 32 * (double)NACC * iters * 4.0 * cus * blocks_per_cu;
        printf("%s, %d wave(s)/SIMD: %.3f ms  %.0f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms / 1e9);
    }
}

int main() {
    int cus = 256;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d;
    hipMalloc(&d, 4096);
    run<16>("16 independent accumulators", 1, cus, d);
    run<16>("16 independent accumulators", 2, cus, d);
    run<4>("4 independent accumulators", 2, cus, d);
    return 0;
}
