// Does the MFMA shape matter for a GEMM-like inner loop at this part's power-limited clock? (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe && tools/bin/mfma_shape_probe
// One 512-thread workgroup per CU (2 waves per SIMD), every wave computes a 128 x 64 accumulator tile from fragments it
// reads out of LDS (random bf16 data, conflict-free lane-linear ds_read_b128, 12 reads per 32-wide K step — the byte and
// flop counts of gemm_pp.hip's main loop, without DMA, barriers or epilogue):
//   SHAPE 0: 32 x v_mfma_f32_16x16x32_bf16 per K step (what gemm_pp.hip issues)
//   SHAPE 1: 16 x v_mfma_f32_32x32x16_bf16 per K step (half the instructions and half the operand-register reads per flop)
// Prints TFLOP/s and the shader clock (s_memtime span of workgroup 0 / wall time).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

template <int SHAPE>
__global__ __launch_bounds__(512) void loop(const uint4* __restrict__ rnd, float* out, unsigned long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 64 KiB of random fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 512) reinterpret_cast<uint4*>(smem)[i] = rnd[(blockIdx.x * 4096 + i) & 0xffff];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const char* base = smem + lane * 16;
    float sum = 0.f;
    if (SHAPE == 0) {
        f32x4 acc[8][4];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
            const char* p = base + ((it + wave) & 3) * 16384;
            bf16x8 a[8], b[4];
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = *reinterpret_cast<const bf16x8*>(p + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j) b[j] = *reinterpret_cast<const bf16x8*>(p + (8 + j) * 1024);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    } else {
        f32x16 acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
            const char* p = base + ((it + wave) & 3) * 16384;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 a[4], b[2];
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *reinterpret_cast<const bf16x8*>(p + (kk * 6 + i) * 1024);
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const bf16x8*>(p + (kk * 6 + 4 + j) * 1024);
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[j], a[i], acc[i][j], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (sum == 1.2345e30f) out[tid] = sum;
    if (blockIdx.x == 0 && tid == 0) *ticks = t1 - t0;
}

template <int SHAPE>
static void run(const char* name, int cus, const uint4* rnd, float* d, unsigned long long* ticks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&loop<SHAPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    loop<SHAPE><<<cus, 512, 65536>>>(rnd, d, ticks, 2000);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        loop<SHAPE><<<cus, 512, 65536>>>(rnd, d, ticks, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t = 0;
        (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        const double flop = 2.0 * 128 * 64 * 32 * (double)iters * 8.0 * cus;
        printf("%-22s %.3f ms  %7.0f TFLOP/s   clock %.2f GHz   %.0f cycles per K step per wave (matrix pipe: 512 per SIMD pair)\n",
               name, ms, flop / ms / 1e9, (double)t / (ms * 1e6), (double)t / iters);
    }
}

int main() {
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    uint4* rnd;
    float* d;
    unsigned long long* ticks;
    (void)hipMalloc(&rnd, 65536 * 16);
    (void)hipMalloc(&d, 4096);
    (void)hipMalloc(&ticks, 8);
    unsigned short* h = (unsigned short*)malloc(65536 * 16);
    srand(1);
    for (int i = 0; i < 65536 * 8; ++i) {        // random bf16 in about [-2, 2]
        const float f = ((rand() & 0xffff) / 32768.0f - 1.0f) * 2.0f;
        unsigned u;
        __builtin_memcpy(&u, &f, 4);
        h[i] = (unsigned short)(u >> 16);
    }
    (void)hipMemcpy(rnd, h, 65536 * 16, hipMemcpyHostToDevice);
    for (int r = 0; r < 2; ++r) {
        run<0>("16x16x32, random data", cus, rnd, d, ticks);
        run<1>("32x32x16, random data", cus, rnd, d, ticks);
    }
    (void)hipMemset(rnd, 0, 65536 * 16);
    run<0>("16x16x32, zeros", cus, rnd, d, ticks);
    run<1>("32x32x16, zeros", cus, rnd, d, ticks);
    return 0;
}
