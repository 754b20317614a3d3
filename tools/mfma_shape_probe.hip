// Does the MFMA shape matter for a GEMM-like inner loop at this part's power-limited clock? (diagnostic)
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_shape_probe.hip -o tools/bin/mfma_shape_probe && tools/bin/mfma_shape_probe
// One 512-thread workgroup per CU (2 waves per SIMD), every wave computes a 128 x 64 accumulator tile from fragments it
// reads out of LDS (random bf16 data, conflict-free lane-linear ds_read_b128, 12 reads per 32-wide K step — the byte and
// flop counts of gemm_pp.hip's main loop, without DMA, barriers or epilogue):
//   SHAPE 0: 32 x v_mfma_f32_16x16x32_bf16 per K step (what gemm_pp.hip issues)
//   SHAPE 1: 16 x v_mfma_f32_32x32x16_bf16 per K step (half the instructions and half the operand-register reads per flop)
//   SHAPE 2 (round 5): 256-thread workgroup, ONE wave per SIMD, 128 x 128 accumulators per wave (256 AccVGPRs), 16 reads and
//            64 x v_mfma_f32_16x16x32_bf16 per K step: 0.25 KiB of LDS reads per MFMA instead of 0.375 — the ceiling of a
//            4-wave 256 x 256 tile kernel, the only structure that lowers the main loop's LDS bytes per flop
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

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void loop_w128(const uint4* __restrict__ rnd, float* out, unsigned long long* ticks, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 64 KiB of random fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) reinterpret_cast<uint4*>(smem)[i] = rnd[(blockIdx.x * 4096 + i) & 0xffff];
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const char* base = smem + lane * 16;
    float sum = 0.f;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // software-pipelined by hand: the fragments of K step it + 1 are read while the 64 MFMAs of step it issue (a lone wave
    // has nobody else to hide its LDS latency behind)
    bf16x8 a0[8], b0[8], a1[8], b1[8];
#define LOADF(A_, B_, P_)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) A_[i] = *reinterpret_cast<const bf16x8*>((P_) + i * 1024);     \
    _Pragma("unroll") for (int j = 0; j < 8; ++j) B_[j] = *reinterpret_cast<const bf16x8*>((P_) + (8 + j) * 1024);
#define MMAF(A_, B_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                \
        _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                            \
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(B_[j], A_[i], acc[i][j], 0, 0, 0);
    LOADF(a0, b0, base + (wave & 3) * 16384)
    for (int it = 0; it < iters; it += 2) {
        const char* p1 = base + ((it + 1 + wave) & 3) * 16384;
        const char* p2 = base + ((it + 2 + wave) & 3) * 16384;
        LOADF(a1, b1, p1)
        MMAF(a0, b0)
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        LOADF(a0, b0, p2)
        MMAF(a1, b1)
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (sum == 1.2345e30f) out[tid] = sum;
    if (blockIdx.x == 0 && tid == 0) *ticks = t1 - t0;
}

// SHAPE 3: SHAPE 2 with the register files chosen by hand (what attention_v4.hip had to do as well): left to the compiler the
// loop above carries 260 v_accvgpr_write + 64 v_accvgpr_mov per 128 MFMAs. Accumulators = AccVGPRs a[0:255], owned by the
// asm statements; fragments in arch VGPRs, read with asm ds_reads in program order, 1 read behind every 4 MFMAs.
template <int I, int J>
__device__ __forceinline__ void mf3(const bf16x8& b, const bf16x8& a) {
    asm volatile("v_mfma_f32_16x16x32_bf16 a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(b), "v"(a), "i"((I * 8 + J) * 4),
                 "i"((I * 8 + J) * 4 + 3));
}
template <int OFF>
__device__ __forceinline__ void rd3(bf16x8& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF));
}
template <int I>
__device__ __forceinline__ void row3(const bf16x8 (&b)[8], const bf16x8& a) {
    mf3<I, 0>(b[0], a); mf3<I, 1>(b[1], a); mf3<I, 2>(b[2], a); mf3<I, 3>(b[3], a);
    mf3<I, 4>(b[4], a); mf3<I, 5>(b[5], a); mf3<I, 6>(b[6], a); mf3<I, 7>(b[7], a);
}
template <int I>
__device__ __forceinline__ void half3a(const bf16x8 (&b)[8], const bf16x8& a) {
    mf3<I, 0>(b[0], a); mf3<I, 1>(b[1], a); mf3<I, 2>(b[2], a); mf3<I, 3>(b[3], a);
}
template <int I>
__device__ __forceinline__ void half3b(const bf16x8 (&b)[8], const bf16x8& a) {
    mf3<I, 4>(b[4], a); mf3<I, 5>(b[5], a); mf3<I, 6>(b[6], a); mf3<I, 7>(b[7], a);
}
// one K step: 64 MFMAs on (Ac, Bc) with the 16 fragment reads of the next step (An, Bn) placed one behind every 4 MFMAs
#define P3_ROW(I, Ac, Bc, RD0, RD1) half3a<I>(Bc, Ac[I]); RD0; half3b<I>(Bc, Ac[I]); RD1;
#define P3_KSTEP(Ac, Bc, An, Bn, AD)                                              \
    P3_ROW(0, Ac, Bc, rd3<0 * 1024>(An[0], AD), rd3<8 * 1024>(Bn[0], AD))         \
    P3_ROW(1, Ac, Bc, rd3<1 * 1024>(An[1], AD), rd3<9 * 1024>(Bn[1], AD))         \
    P3_ROW(2, Ac, Bc, rd3<2 * 1024>(An[2], AD), rd3<10 * 1024>(Bn[2], AD))        \
    P3_ROW(3, Ac, Bc, rd3<3 * 1024>(An[3], AD), rd3<11 * 1024>(Bn[3], AD))        \
    P3_ROW(4, Ac, Bc, rd3<4 * 1024>(An[4], AD), rd3<12 * 1024>(Bn[4], AD))        \
    P3_ROW(5, Ac, Bc, rd3<5 * 1024>(An[5], AD), rd3<13 * 1024>(Bn[5], AD))        \
    P3_ROW(6, Ac, Bc, rd3<6 * 1024>(An[6], AD), rd3<14 * 1024>(Bn[6], AD))        \
    P3_ROW(7, Ac, Bc, rd3<7 * 1024>(An[7], AD), rd3<15 * 1024>(Bn[7], AD))        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void loop_w128_asm(const uint4* __restrict__ rnd, float* out, unsigned long long* ticks, int iters) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];     // 64 KiB of random fragments
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 4096; i += 256) reinterpret_cast<uint4*>(smem)[i] = rnd[(blockIdx.x * 4096 + i) & 0xffff];
    __syncthreads();
    // zero the accumulators; the clobber list makes the kernel allocate the whole accumulator file
    asm volatile(
        ".set i, 0\n\t.rept 256\n\tv_accvgpr_write_b32 a[i], 0\n\t.set i, i + 1\n\t.endr" ::: "a0", "a63", "a64", "a127", "a128",
        "a191", "a192", "a255");
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const unsigned base = (unsigned)(uintptr_t)(smem) + lane * 16;
    bf16x8 a0[8], b0[8], a1[8], b1[8];
    {
        const unsigned ad = base + (wave & 3) * 16384;
        rd3<0>(a0[0], ad); rd3<1024>(a0[1], ad); rd3<2048>(a0[2], ad); rd3<3072>(a0[3], ad);
        rd3<4096>(a0[4], ad); rd3<5120>(a0[5], ad); rd3<6144>(a0[6], ad); rd3<7168>(a0[7], ad);
        rd3<8192>(b0[0], ad); rd3<9216>(b0[1], ad); rd3<10240>(b0[2], ad); rd3<11264>(b0[3], ad);
        rd3<12288>(b0[4], ad); rd3<13312>(b0[5], ad); rd3<14336>(b0[6], ad); rd3<15360>(b0[7], ad);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma nounroll
    for (int it = 0; it < iters; it += 2) {
        const unsigned p1 = base + ((it + 1 + wave) & 3) * 16384;
        const unsigned p2 = base + ((it + 2 + wave) & 3) * 16384;
        P3_KSTEP(a0, b0, a1, b1, p1)
        P3_KSTEP(a1, b1, a0, b0, p2)
    }
    float sum;
    asm volatile("s_nop 7\n\ts_nop 7\n\tv_accvgpr_read_b32 %0, a0" : "=v"(sum));
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (sum == 1.2345e30f) out[tid] = sum;
    if (blockIdx.x == 0 && tid == 0) *ticks = t1 - t0;
#endif
}

template <typename KF>
static void run_w128_(KF kernel, const char* name, int cus, const uint4* rnd, float* d, unsigned long long* ticks) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    kernel<<<cus, 256, 65536>>>(rnd, d, ticks, 2000);
    (void)hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        (void)hipEventRecord(e0);
        kernel<<<cus, 256, 65536>>>(rnd, d, ticks, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long t = 0;
        (void)hipMemcpy(&t, ticks, 8, hipMemcpyDeviceToHost);
        const double flop = 2.0 * 128 * 128 * 32 * (double)iters * 4.0 * cus;
        printf("%-22s %.3f ms  %7.0f TFLOP/s   clock %.2f GHz   %.0f ticks per K step per wave (64 MFMAs, one wave per SIMD)\n",
               name, ms, flop / ms / 1e9, (double)t / (ms * 1e6), (double)t / iters);
    }
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
        run_w128_(loop_w128, "128x128/wave, random", cus, rnd, d, ticks);
        run_w128_(loop_w128_asm, "128x128/wave asm, rnd", cus, rnd, d, ticks);
    }
    (void)hipMemset(rnd, 0, 65536 * 16);
    run<0>("16x16x32, zeros", cus, rnd, d, ticks);
    run<1>("32x32x16, zeros", cus, rnd, d, ticks);
    run_w128_(loop_w128, "128x128/wave, zeros", cus, rnd, d, ticks);
    run_w128_(loop_w128_asm, "128x128/wave asm, 0s", cus, rnd, d, ticks);
    return 0;
}
