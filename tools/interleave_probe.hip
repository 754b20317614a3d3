// Can a wave hide its OWN VALU work in the shadow of its OWN MFMAs?  (diagnostic, not part of the library)
// tools/coissue_probe.hip showed that a VALU-only wave makes no progress next to an MFMA-only wave of the same SIMD (the
// older wave's stalled MFMA holds the VALU issue port). This probe asks the in-wave question: every wave runs
//      16 x { v_mfma_f32_32x32x16_bf16 (four independent accumulators in rotation) ; NV independent VALU instructions }
// per iteration, written as asm volatile so that the order in the binary is the order here, with W = 1 / 2 / 3 waves per SIMD.
//   cycles per iteration == 512                       -> the VALU instructions ride in the MFMA's 32-cycle shadow for free
//   cycles per iteration == 512 + 16 * NV * issue     -> they serialise, as they do across waves
//   hipcc --offload-arch=gfx950 -O3 tools/interleave_probe.hip -o /tmp/interleave_probe && /tmp/interleave_probe
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

#define MFMA(ACC) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(ACC) : "v"(a), "v"(b))
#define VFMA(X) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(X) : "v"(c0), "v"(c1))
#define VEXP(X) asm volatile("v_exp_f32 %0, %0" : "+v"(X))
#define VCVT(D, X, Y) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(D) : "v"(X), "v"(Y))

// KIND 0: NV x v_fma_f32; 1: NV x v_exp_f32; 2: the softmax mix per MFMA (2 exp + 2 add(fma) + 1 cvt_pk) x NV
template <int NV, int KIND>
__global__ void interleave(float* out, int iters, unsigned long long* ticks) {
    f32x16 acc0, acc1, acc2, acc3;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = acc2[r] = acc3[r] = 0.f;
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
    float x[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) x[k] = (float)(threadIdx.x + k) * 1e-3f;
    float c0 = 0.999f, c1 = 1e-3f;
    unsigned pk = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if ((u & 3) == 0) MFMA(acc0);
            if ((u & 3) == 1) MFMA(acc1);
            if ((u & 3) == 2) MFMA(acc2);
            if ((u & 3) == 3) MFMA(acc3);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                if (KIND == 0) VFMA(x[v & 7]);
                if (KIND == 1) VEXP(x[v & 7]);
                if (KIND == 2) {
                    VEXP(x[(2 * v) & 7]);
                    VEXP(x[(2 * v + 1) & 7]);
                    VFMA(x[(2 * v + 4) & 7]);
                    VFMA(x[(2 * v + 5) & 7]);
                    unsigned t;
                    VCVT(t, x[(2 * v + 2) & 7], x[(2 * v + 3) & 7]);
                    pk ^= t;                                   // one more VALU (v_xor) — counted in the table below
                }
            }
        }
    }
    asm volatile("s_nop 15\n s_nop 15\n s_nop 15" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float sink = acc0[0] + acc1[3] + acc2[7] + acc3[11] + (float)pk;
#pragma unroll
    for (int k = 0; k < 8; ++k) sink += x[k];
    if (blockIdx.x == 0 && (threadIdx.x & 63) == 0) ticks[threadIdx.x >> 6] = t1 - t0;
    if (sink == 1.2345e30f) out[threadIdx.x] = sink;
}

template <int NV, int KIND>
static void run(const char* name, int cus, int waves_per_simd, float* d, unsigned long long* ticks) {
    const int iters = 20000;
    const int threads = 256 * waves_per_simd;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((interleave<NV, KIND>), dim3(cus), dim3(threads), 0, 0, d, 100, ticks);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((interleave<NV, KIND>), dim3(cus), dim3(threads), 0, 0, d, iters, ticks);
    (void)hipEventRecord(e1, 0);
    (void)hipDeviceSynchronize();
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[16];
    (void)hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
    const double mf = 4.0 * waves_per_simd * cus * (double)iters * 16 * 32768.0 / (ms * 1e-3) / 1e12;
    // SIMD cycles per wave-iteration: the wave's own span / waves per SIMD
    printf("%-40s W=%d  %8.3f ms  %7.1f TFLOP/s   memtime ticks / iter / wave-of-SIMD %7.1f\n", name, waves_per_simd, ms, mf,
           (double)h[0] / iters / waves_per_simd);
}

template <int NV, int KIND>
static void sweep(const char* name, int cus, float* d, unsigned long long* ticks) {
    for (int w = 1; w <= 3; ++w) run<NV, KIND>(name, cus, w, d, ticks);
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    float* d;
    unsigned long long* ticks;
    (void)hipMalloc(&d, 8192);
    (void)hipMalloc(&ticks, 128);
    sweep<0, 0>("16 MFMA", cus, d, ticks);
    sweep<2, 0>("16 x (MFMA + 2 fma)", cus, d, ticks);
    sweep<4, 0>("16 x (MFMA + 4 fma)", cus, d, ticks);
    sweep<6, 0>("16 x (MFMA + 6 fma)", cus, d, ticks);
    sweep<8, 0>("16 x (MFMA + 8 fma)", cus, d, ticks);
    sweep<1, 1>("16 x (MFMA + 1 exp)", cus, d, ticks);
    sweep<2, 1>("16 x (MFMA + 2 exp)", cus, d, ticks);
    sweep<3, 1>("16 x (MFMA + 3 exp)", cus, d, ticks);
    sweep<4, 1>("16 x (MFMA + 4 exp)", cus, d, ticks);
    sweep<1, 2>("16 x (MFMA + 2 exp 2 fma cvt xor)", cus, d, ticks);
    return 0;
}
