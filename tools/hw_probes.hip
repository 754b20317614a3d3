// Sustained MFMA throughput and LDS read-rate probes (diagnostic, not part of the library): every wave issues independent
// v_mfma_f32_16x16x32_bf16 back to back from registers only — the ceiling any GEMM main loop on this part sits under at
// the clocks the chip actually holds.   hipcc --offload-arch=gfx950 -O3 tools/hw_probes.hip -o /tmp/hw_probes && /tmp/hw_probes
#include <hip/hip_runtime.h>
#include <stdio.h>

using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int NACC, int UNR>
__global__ __launch_bounds__(256) void mfma_loop(float* out, int iters) {
    f32x4 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
    for (int it = 0; it < iters; it += UNR) {
#pragma unroll
        for (int u = 0; u < UNR; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e30f) out[threadIdx.x] = s;
}

// LDS read rate: every wave streams conflict-free ds_read_b128 (lane-linear 1 KiB per instruction) out of a 64 KiB
// region; bytes per clock per CU = what bounds the fragment reads of an LDS-staged GEMM.
template <int PAT>
__global__ __launch_bounds__(512) void lds_read_loop(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    const int tid = threadIdx.x;
    for (int i = tid; i < 16384; i += blockDim.x) reinterpret_cast<unsigned int*>(smem)[i] = i;
    __syncthreads();
    u32x4 acc = {0, 0, 0, 0};
    // PAT 0: lane-linear; 1: MFMA A-fragment rows (row = lane & 15, 128-B rows), chunk (lane>>4) ^ (row & 7);
    // 2: same rows, chunk (lane>>4) ^ ((row >> 1) & 7)
    const int lane = tid & 63, frow = lane & 15, fq = lane >> 4;
    const int off = PAT == 0 ? lane * 16 : frow * 128 + ((fq ^ (PAT == 1 ? (frow & 7) : ((frow >> 1) & 7))) << 4);
    const char* base = smem + off + (tid >> 6) * 8192;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(base + (PAT == 0 ? k * 1024 : (k >> 1) * 2048 + (k & 1) * 64));
            acc ^= v;
        }
        asm volatile("" ::: "memory");
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[tid] = 1.f;
}

template <int PAT>
static void run_lds(int waves, int cus, float* d) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&lds_read_loop<PAT>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    lds_read_loop<PAT><<<cus, waves * 64, 65536>>>(d, 16);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    lds_read_loop<PAT><<<cus, waves * 64, 65536>>>(d, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)cus * waves * iters * 8 * 1024.0;
    printf("ds_read_b128 pattern %d, %d waves per CU: %.3f ms  %.1f TB/s chip-wide = %.1f B/clk/CU at 2.1 GHz\n", PAT, waves, ms,
           bytes / ms / 1e9, bytes / (ms * 1e-3) / cus / 2.1e9);
}


// Interference probe: waves 0-3 (one per SIMD) stream independent MFMAs, waves 4-7 (their SIMD partners) run a side
// stream until the MFMA waves are done: SIDE 0 nothing, 1 `buffer_load_dwordx4 ... lds` (1 KiB per instruction) from an
// L2-resident buffer, 2 the same loads into registers, 3 conflict-free ds_read_b128, 4 DMA throttled by s_sleep.
// Reports the MFMA rate and the side stream's rate: what a ping-pong GEMM's load row costs its MFMA row.
#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
template <int SIDE>
__global__ __launch_bounds__(512) void mixed_loop(float* out, const char* src, unsigned long long* side_ops, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    volatile int* flag = reinterpret_cast<volatile int*>(smem + 65536);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid == 0) *flag = 0;
    __syncthreads();
    if (wave < 4) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 a = {1, 2, 3, 4, 5, 6, 7, (short)threadIdx.x}, b = {8, 7, 6, 5, 4, 3, 2, (short)blockIdx.x};
        __builtin_amdgcn_s_setprio(1);
        for (int it = 0; it < iters; it += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) sum += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        if (sum == 1.2345e30f) out[threadIdx.x] = sum;
        if (wave == 0 && lane == 0) *flag = 1;
        return;
    }
    unsigned long long n = 0;
    if (SIDE == 0) return;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 1 << 20, 0x00020000);
    const int voff = ((blockIdx.x & 15) << 16) + (wave - 4) * 8192 + lane * 16;
    char* dst = smem + (wave - 4) * 16384;
    u32x4 acc = {0, 0, 0, 0};
    while (*flag == 0) {
        if (SIDE == 1 || SIDE == 4) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, LDS_AS(dst + k * 1024), 16, voff + k * 1024, 0, 0, 0);
                if (SIDE == 4) __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else if (SIDE == 2) {
#pragma unroll
            for (int k = 0; k < 8; ++k)
                acc ^= __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(src + voff + k * 1024));
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) acc ^= *reinterpret_cast<const u32x4*>(dst + lane * 16 + k * 1024);
            asm volatile("" ::: "memory");
        }
        n += 8;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[tid] = 1.f;
    if (lane == 0) atomicAdd(side_ops, n);
}

template <int SIDE>
static void run_mixed(const char* name, int cus, float* d, const char* src, unsigned long long* ops) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mixed_loop<SIDE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536 + 64);
    mixed_loop<SIDE><<<cus, 512, 65536 + 64>>>(d, src, ops, 64);
    (void)hipDeviceSynchronize();
    (void)hipMemset(ops, 0, 8);
    (void)hipEventRecord(e0);
    mixed_loop<SIDE><<<cus, 512, 65536 + 64>>>(d, src, ops, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long n = 0;
    (void)hipMemcpy(&n, ops, 8, hipMemcpyDeviceToHost);
    const double flop = 2.0 * 16 * 16 * 32 * 16.0 * iters * 4.0 * cus;
    printf("MFMA row + side stream [%s]: %.3f ms  MFMA %.0f TFLOP/s   side %.1f B/clk/CU at 2.1 GHz (%.2f KiB per 16 MFMAs per CU)\n",
           name, ms, flop / ms / 1e9, (double)n * 1024.0 / (ms * 1e-3) / cus / 2.1e9,
           (double)n / cus / (iters / 1.0));
}

// Store-rate probe: one 512-thread block per CU, every wave streams 16-byte-per-lane stores (1 KiB per instruction,
// 2 rows x 512 contiguous bytes like the GEMM epilogue) into its own region. MODE 0 plain, 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt;
// `span` bytes per CU are cycled (span = 128 KiB: L2-resident target; large: streams to HBM).
template <int MODE>
__global__ __launch_bounds__(512) void store_loop(char* dst, size_t span, int iters) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    const int tid = threadIdx.x;
    char* base = dst + (size_t)blockIdx.x * span;
    u32x4 v = {1u, 2u, 3u, (unsigned)tid};
    size_t off = (size_t)tid * 16;
    for (int it = 0; it < iters; ++it) {
        char* ptr = base + off;
        if (MODE == 0) asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(ptr), "v"(v) : "memory");
        if (MODE == 1) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(ptr), "v"(v) : "memory");
        if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(ptr), "v"(v) : "memory");
        if (MODE == 3) asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1 nt" ::"v"(ptr), "v"(v) : "memory");
        off += 8192;
        if (off >= span) off = (size_t)tid * 16;
    }
}

template <int MODE>
static void run_store(const char* name, int cus, int blocks, char* buf, size_t span) {
    const int iters = 2048;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    store_loop<MODE><<<blocks, 512>>>(buf, span, 64);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    store_loop<MODE><<<blocks, 512>>>(buf, span, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * iters * 8192.0;
    printf("stores [%s], %d blocks, %zu KiB per block: %.3f ms  %.2f TB/s chip-wide = %.1f GB/s per block\n", name, blocks,
           span >> 10, ms, bytes / ms / 1e9, bytes / ms / 1e6 / blocks);
}

// The GEMM epilogue's store pattern in isolation: persistent blocks walk 256x256 bf16 tiles of a row-major M x N matrix
// in the kernel's tile order (8 m-tiles x all n-tiles per group, XCD-contiguous), each wave instruction writes 2 rows x
// 512 B. ORDER 0: the kernel's order; 1: n-fastest inside a group; 2: kernel order, per-wave 128 x 64 strips (8 rows x 128 B).
template <int ORDER>
__global__ __launch_bounds__(512) void tile_store_loop(char* dst, int M, int N, int reps) {
    using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
    const int tid = threadIdx.x, cg = tid & 31, r0 = tid >> 5;
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256, total = tiles_m * tiles_n;
    const u32x4 val = {1u, 2u, 3u, (unsigned)tid};
    for (int rep = 0; rep < reps; ++rep)
        for (int v = blockIdx.x; v < total; v += gridDim.x) {
            const int q = total >> 3, r = total & 7, xcd = v & 7;
            const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
            const int gsz = 8 * tiles_n, g = wg / gsz, first_m = g * 8;
            const int gm = min(tiles_m - first_m, 8), in = wg - g * gsz;
            const int tm = ORDER != 1 ? first_m + in % gm : first_m + in / tiles_n;
            const int tn = ORDER != 1 ? in / gm : in % tiles_n;
            if (ORDER == 2) {        // per-wave 128 x 64 strips: one instruction = 8 rows x 128 B
                const int lane = tid & 63, wave = tid >> 6, wm = wave >> 2, wn = wave & 3;
                for (int i = 0; i < 8; ++i)
                    for (int t = 0; t < 2; ++t) {
                        const int m = tm * 256 + wm * 128 + i * 16 + t * 8 + (lane >> 3);
                        if (m < M) {
                            char* ptr = dst + ((size_t)m * N + tn * 256 + wn * 64 + (lane & 7) * 8) * 2;
                            asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(ptr), "v"(val) : "memory");
                        }
                    }
                continue;
            }
            for (int c = 0; c < 4; ++c)
                for (int k = 0; k < 4; ++k) {
                    const int m = tm * 256 + c * 64 + k * 16 + r0;
                    if (m < M) {
                        char* ptr = dst + ((size_t)m * N + tn * 256 + cg * 8) * 2;
                        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(ptr), "v"(val) : "memory");
                    }
                }
        }
}

template <int ORDER>
static void run_tile_store(int cus, char* buf, int M, int N) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    tile_store_loop<ORDER><<<cus, 512>>>(buf, M, N, 1);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    tile_store_loop<ORDER><<<cus, 512>>>(buf, M, N, 4);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double bytes = 4.0 * M * N * 2.0;
    printf("tile stores (GEMM epilogue pattern, order %d) M=%d N=%d: %.3f ms per pass  %.2f TB/s = %.1f GB/s per CU\n", ORDER, M, N,
           ms / 4, bytes / ms / 1e9, bytes / ms / 1e6 / cus);
}

template <int NACC, int UNR>
static void run(const char* name, int blocks_per_cu, int cus, float* d) {
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    mfma_loop<NACC, UNR><<<cus * blocks_per_cu, 256>>>(d, 64);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        mfma_loop<NACC, UNR><<<cus * blocks_per_cu, 256>>>(d, iters);
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
    run<16, 1>("16 independent accumulators", 1, cus, d);
    run<16, 8>("16 independent accumulators, loop unrolled x8", 1, cus, d);
    run<16, 1>("16 independent accumulators", 2, cus, d);
    run<16, 8>("16 independent accumulators, loop unrolled x8", 2, cus, d);
    run<4, 8>("4 independent accumulators, unrolled x8", 2, cus, d);
    char* src;
    unsigned long long* ops;
    hipMalloc(&src, 1 << 20);
    hipMemset(src, 1, 1 << 20);
    hipMalloc(&ops, 8);
    run_mixed<0>("none", cus, d, src, ops);
    run_mixed<1>("buffer_load ... lds", cus, d, src, ops);
    run_mixed<4>("buffer_load ... lds, throttled", cus, d, src, ops);
    run_mixed<2>("global_load_dwordx4 to registers", cus, d, src, ops);
    run_mixed<3>("ds_read_b128", cus, d, src, ops);
    {
        char* big;
        const size_t span_hbm = 16u << 20;
        hipMalloc(&big, (size_t)cus * span_hbm);
        for (int blocks : {cus, cus / 2, cus / 8, 8}) {
            run_store<0>("plain, HBM", cus, blocks, big, span_hbm);
            run_store<1>("nt, HBM", cus, blocks, big, span_hbm);
            run_store<2>("sc0 sc1, HBM", cus, blocks, big, span_hbm);
            run_store<3>("sc0 sc1 nt, HBM", cus, blocks, big, span_hbm);
        }
        run_store<0>("plain, L2-resident", cus, cus, big, 128u << 10);
        run_store<1>("nt, L2-resident", cus, cus, big, 128u << 10);
        for (int N : {1024, 3072, 4096, 2048}) {
            run_tile_store<0>(cus, big, 139400, N);
            run_tile_store<1>(cus, big, 139400, N);
            run_tile_store<2>(cus, big, 139400, N);
        }
        hipFree(big);
    }
    run_lds<0>(8, cus, d);
    run_lds<1>(8, cus, d);
    run_lds<2>(8, cus, d);
    return 0;
}
