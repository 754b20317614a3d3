// bf16 "lone-wave" tile GEMM: 256 x 256 x 64 tiles, FOUR waves — one per SIMD — each owning a 128 x 128 accumulator tile in
// the 256 AccVGPRs of its SIMD lane file, ONE persistent workgroup per CU, v_mfma_f32_16x16x32_bf16.
//
// Why a second frame beside gemm_pp.hip (8 waves, 128 x 64 per wave): at this part's power-limited clock the matrix pipes of the
// ping-pong kernel's main loop are ~0.9 busy — what is left is energy per flop. A 128 x 128 wave tile reads 16 fragments per 64
// MFMAs (0.25 KiB of LDS per MFMA) where 128 x 64 reads 12 per 32 (0.375): profiles/r5_mfma_shape_probe.txt measured the
// LDS-fed loops at 1.70 (128 x 64, two waves per SIMD) against 1.84 - 1.89 PFLOP/s (this frame). A wave that is alone on its SIMD
// has nobody to hide its waits behind, so every wait of the main loop is placed where its condition already holds:
//
//   K tile = 2 k-steps of 64 MFMAs. The 16 fragment reads of the NEXT k-step are issued behind the first 44 MFMAs of the current
//     one (3 reads per 8 MFMAs), so the `s_waitcnt lgkmcnt(0)` at its end waits for reads issued >= 300 matrix cycles earlier.
//   staging: 2 stages of 64 KiB, `buffer_load_dwordx4 ... lds` with the source-side XOR swizzle of gemm_pp.hip (same LDS image,
//     same fragment addresses); a wave issues the 16 1-KiB DMAs of K tile g + 2 behind the MFMAs of K tile g's second k-step.
//   ONE workgroup barrier per K tile, between its k-steps: behind it every wave has finished reading stage g & 1 (so the DMAs
//     of K tile g + 2 may overwrite it) and — `s_waitcnt vmcnt` in front of it — K tile g + 1 has landed (so the second k-step
//     may read its first fragments). The K-tile sequence g runs ACROSS output tiles: the first two K tiles of tile t + 1 are in
//     flight before tile t's epilogue issues its first store (the epilogue stages through 16 KiB of LDS of its own, behind the
//     two stages), and the wait for K tile 1 counts the epilogue's stores out (vmcnt retires in order).
//   accumulators: a[0:255], named only by the inline asm of this file (MFMAs, the read-out of the epilogue, the bias start
//     values); the compiler owns nothing there (tools/gemm_lw/lw_audit.py audits the ISA). The first k-step of an output tile issues
//     its MFMAs with C = 0 — no zeroing pass.
//   epilogue: the wave's 128 x 128 strip is two 128 x 64 strips; each is read out of the AccVGPRs (v_accvgpr_read) into the
//     registers the fragments used and handed to gemm_pp.hip's per-wave epilogue (gemm_epilogue_wave.inc: the same code, the
//     same roundings — outputs are bit-identical between the two kernels).
#include <stdlib.h>
#include <type_traits>
#include <utility>

#include "gemm_epilogue.h"      // (csrc/: tools/build_variant.sh compiles this file with -I csrc)

#define LBM 256
#define LBK 64
#define LHALF (128 * 128)             // 16 KiB: 128 rows x 64 bf16
#define LSTAGE (4 * LHALF)            // [A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255]
#define LW_EPI_BYTES (4 * 4096)       // per-wave epilogue staging (4 KiB each), behind the two stages
#define PHALF LHALF                   // (names the shared epilogue uses)

#ifndef PP_ST_AUX
#define PP_ST_AUX 2
#endif
#ifndef PP_RES_AUX
#define PP_RES_AUX 2
#endif
#ifndef PP_NT_PTR_STORES
#define PP_NT_PTR_STORES 1
#endif
#ifdef LW_NOSTORE      /* diagnostic build: main loop only */
#define LW_DIAG_NOSTORE true
#else
#define LW_DIAG_NOSTORE false
#endif
#define PP_DIAG_L2STORE false

#if GAR_HALF_F16
#define LW_MFMA "v_mfma_f32_16x16x32_f16"
#else
#define LW_MFMA "v_mfma_f32_16x16x32_bf16"
#endif

// 16-byte output stores of the pointer-addressed epilogues (shared epilogue code)
__device__ __forceinline__ void pp_st8(bf16_t* p, const float (&v)[8]) {
    const u32x4 w = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    if (PP_NT_PTR_STORES) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = w;
}
// (BIAS_GELU is not served by this kernel: its GEMMs have K = 1024; the shared epilogue refers to the table lookup)
__device__ __forceinline__ float gelu_lut(float x, const char*) { return x; }

// acc tile (I, J) = a[(I * 8 + J) * 4 .. + 3]:  acc += Wfrag[J] (A operand: rows = n) x Afrag[I] (B operand: columns = m)
template <int R, bool ZERO>
__device__ __forceinline__ void lw_mma(const bf16x8& w, const bf16x8& a) {
    if (ZERO) asm volatile(LW_MFMA " a[%c2:%c3], %0, %1, 0" ::"v"(w), "v"(a), "i"(R), "i"(R + 3));
    else asm volatile(LW_MFMA " a[%c2:%c3], %0, %1, a[%c2:%c3]" ::"v"(w), "v"(a), "i"(R), "i"(R + 3));
}
template <int OFF>
__device__ __forceinline__ void lw_rd(bf16x8& f, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f) : "v"(addr), "i"(OFF));
}
// four accumulators of tile (I, J) -> arch VGPRs
template <int R>
__device__ __forceinline__ void lw_acc_get(f32x4& d) {
    float x0, x1, x2, x3;
    asm volatile("v_accvgpr_read_b32 %0, a%c4\n\tv_accvgpr_read_b32 %1, a%c5\n\tv_accvgpr_read_b32 %2, a%c6\n\tv_accvgpr_read_b32 %3, a%c7"
                 : "=v"(x0), "=v"(x1), "=v"(x2), "=v"(x3)
                 : "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
    d = f32x4{x0, x1, x2, x3};
}
template <int R>
__device__ __forceinline__ void lw_acc_set(const float (&v)[4]) {
    asm volatile("v_accvgpr_write_b32 a%c4, %0\n\tv_accvgpr_write_b32 a%c5, %1\n\tv_accvgpr_write_b32 a%c6, %2\n\tv_accvgpr_write_b32 a%c7, %3"
                 ::"v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "i"(R), "i"(R + 1), "i"(R + 2), "i"(R + 3));
}
template <int H, int... IJ>
__device__ __forceinline__ void lw_acc_copy(f32x4 (&acc)[8][4], std::integer_sequence<int, IJ...>) {
    (lw_acc_get<((IJ / 4) * 8 + H * 4 + (IJ % 4)) * 4>(acc[IJ / 4][IJ % 4]), ...);
}
// every accumulator tile (I, J), I = 0..7, starts from the same four values bv[J]
template <int... IJ>
__device__ __forceinline__ void lw_acc_fill(const float (&bv)[8][4], std::integer_sequence<int, IJ...>) {
    (lw_acc_set<IJ * 4>(bv[IJ % 8]), ...);
}

#ifdef LW_TIMELINE   /* diagnostic build (tools/lw_timeline.py): shader-clock stamps at the k-step boundaries, sums per wave of workgroup 0 */
__device__ __forceinline__ unsigned lw_now() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return (unsigned)t;
}
#define LW_TL(i) { const unsigned t_ = lw_now(); tl_sum[i] += t_ - tl_t; tl_t = t_; }
#else
#define LW_TL(i)
#endif

// W fragment row offsets inside the wave's 128-row W half (see gemm_pp.hip: the rows of an n-tile pair are fed in a permuted
// order so that a lane ends up with eight consecutive output columns; SwiGLU pairs (gate16 | up16) tiles and keeps row order)
template <bool PERM, int J>
struct lw_woff {
    static constexpr int value = PERM ? 8192 * (J >> 2) + 4096 * ((J >> 1) & 1) + 512 * (J & 1) : 2048 * J;
};

#define LW_M4(I, J0, Z, Ac, Bc)                                                                           \
    lw_mma<((I) * 8 + (J0) + 0) * 4, Z>(Bc[(J0) + 0], Ac[I]); lw_mma<((I) * 8 + (J0) + 1) * 4, Z>(Bc[(J0) + 1], Ac[I]); \
    lw_mma<((I) * 8 + (J0) + 2) * 4, Z>(Bc[(J0) + 2], Ac[I]); lw_mma<((I) * 8 + (J0) + 3) * 4, Z>(Bc[(J0) + 3], Ac[I]);
#define LW_RDB(RD, Bn, rb_, J) if (RD) lw_rd<lw_woff<PERM, J>::value>(Bn[J], rb_);
#define LW_RDA(RD, An, ra_, I) if (RD) lw_rd<(I) * 2048>(An[I], ra_);
// DMA instruction q = 0..15 of the prefetch cursor's K tile: half q >> 2 of the stage (A rows 0-127, A rows 128-255, W rows 0-127,
// W rows 128-255), chunk g = q & 3 of the FOUR CONSECUTIVE 1-KiB chunks (rows 32 wave + 8 g ..) this wave stages of every half:
// they share one LDS base (M0 is written once per half, not once per DMA — a lone wave pays for every scalar instruction in front
// of a DMA with matrix cycles) and differ in the instruction offset, which the hardware adds to the LDS AND the global address:
// the lane's global offset vA / vW carries - 1024 g to cancel it. One DMA behind every group of four MFMAs (a pair of DMAs back
// to back costs a lone wave ~40 % more idle matrix cycles than two single ones: tools/lw_issue_probe.hip).
#define LW_DMA(Q)                                                                                                     \
    {                                                                                                                 \
        constexpr int q_ = (Q), h_ = q_ >> 2, g_ = q_ & 3;                                                            \
        if (q_ < 8)                                                                                                   \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_AS(nx + h_ * LHALF + wave * 4096), 16, vA[q_], (int)pf_k, g_ * 1024, 0); \
        else                                                                                                          \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_AS(nx + h_ * LHALF + wave * 4096), 16, vW[q_ - 8], (int)pf_k, g_ * 1024, 0); \
    }
#define LW_DMA1(ON, Q) if (ON) LW_DMA(Q)
// One k-step: 64 MFMAs on the fragments (Ac, Bc); behind them, when RD, the 16 fragment reads of the next k-step (An, Bn; LDS byte
// addresses ra_ / rb_ of this lane's A / W fragment rows at that k-step's 16-byte chunk) — all issued behind the first 44 MFMAs —
// and, when DM, the sixteen DMAs of the prefetch cursor's K tile. Ends with every read landed.
#define LW_KSTEP(Z, RD, DM, Ac, Bc, An, Bn, ra_, rb_)                                                                 \
    LW_M4(0, 0, Z, Ac, Bc) LW_RDB(RD, Bn, rb_, 0) LW_RDB(RD, Bn, rb_, 1) LW_DMA1(DM, 0)                                \
    LW_M4(0, 4, Z, Ac, Bc) LW_RDB(RD, Bn, rb_, 2) LW_DMA1(DM, 1)                                                      \
    LW_M4(1, 0, Z, Ac, Bc) LW_RDB(RD, Bn, rb_, 3) LW_RDB(RD, Bn, rb_, 4) LW_DMA1(DM, 2)                                \
    LW_M4(1, 4, Z, Ac, Bc) LW_RDB(RD, Bn, rb_, 5) LW_DMA1(DM, 3)                                                      \
    LW_M4(2, 0, Z, Ac, Bc) LW_RDB(RD, Bn, rb_, 6) LW_RDB(RD, Bn, rb_, 7) LW_DMA1(DM, 4)                                \
    LW_M4(2, 4, Z, Ac, Bc) LW_RDA(RD, An, ra_, 0) LW_DMA1(DM, 5)                                                      \
    LW_M4(3, 0, Z, Ac, Bc) LW_RDA(RD, An, ra_, 1) LW_RDA(RD, An, ra_, 2) LW_DMA1(DM, 6)                                \
    LW_M4(3, 4, Z, Ac, Bc) LW_RDA(RD, An, ra_, 3) LW_DMA1(DM, 7)                                                      \
    LW_M4(4, 0, Z, Ac, Bc) LW_RDA(RD, An, ra_, 4) LW_RDA(RD, An, ra_, 5) LW_DMA1(DM, 8)                                \
    LW_M4(4, 4, Z, Ac, Bc) LW_RDA(RD, An, ra_, 6) LW_DMA1(DM, 9)                                                      \
    LW_M4(5, 0, Z, Ac, Bc) LW_RDA(RD, An, ra_, 7) LW_DMA1(DM, 10)                                                     \
    LW_M4(5, 4, Z, Ac, Bc) LW_DMA1(DM, 11)                                                                            \
    LW_M4(6, 0, Z, Ac, Bc) LW_DMA1(DM, 12)                                                                            \
    LW_M4(6, 4, Z, Ac, Bc) LW_DMA1(DM, 13)                                                                            \
    LW_M4(7, 0, Z, Ac, Bc) LW_DMA1(DM, 14)                                                                            \
    LW_M4(7, 4, Z, Ac, Bc) LW_DMA1(DM, 15)                                                                            \
    if (RD) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

template <int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void gemm_bf16_lw_kernel(const gar_gemm_params p, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr bool PERM = EPI != GAR_EPI_SWIGLU;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x 64 KiB + 16 KiB of epilogue staging
    const int total = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const char* const glut = smem;                                  // (unused: no GELU instantiation)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn2 = wave & 1;
    int wn = wn2 * 2;                                               // the 64-column strip the shared epilogue works on
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const int frow = lane & 15, fq = lane >> 4;
    const int nt = p.K / LBK;

    // the whole accumulator file belongs to the asm of this kernel (the clobbers make the kernel allocate it)
    asm volatile("" ::: "a0", "a63", "a64", "a127", "a128", "a191", "a192", "a255");

    const unsigned rbA = (unsigned)p.lda * 2u, rbW = (unsigned)p.ldw * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)A, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
    // DMA: a 128-row half = 16 chunks of 8 rows (1 KiB, one wave instruction each); wave w stages chunks 4 w .. 4 w + 3 (rows 32 w ..).
    // Lane l of chunk c: row 8 c + (l >> 3), 16-byte column (l & 7) ^ key(row) — the swizzle keys of gemm_pp.hip:
    //   A halves and SwiGLU W halves: key = row & 7;   W halves otherwise: key = ((row >> 3) & 3) * 2 + ((row >> 1) & 1)
    const int sub = lane >> 3;
    int v = blockIdx.x, tm, tn;
    const int tgm = tile_group_m(p.K, tiles_n);
    tile_of(v, total, tiles_m, tiles_n, tm, tn, tgm);
    int m0 = tm * LBM, n0 = tn * LBM;

    constexpr bool BIAS_INIT = EPI == GAR_EPI_BIAS || EPI == GAR_EPI_BIAS_GELU || EPI == GAR_EPI_BIAS_SCALE_RES ||
                               EPI == GAR_EPI_QKV_ROPE;
    constexpr bool RS_EPI = EPI == GAR_EPI_NONE || EPI == GAR_EPI_BIAS || EPI == GAR_EPI_BIAS_GELU || EPI == GAR_EPI_SWIGLU ||
                            EPI == GAR_EPI_QKV_ROPE || EPI == GAR_EPI_QKV_ROPE_LLM;
    constexpr bool STATS_EPI = EPI == GAR_EPI_RES || EPI == GAR_EPI_BIAS_SCALE_RES;
    const bool RS = RS_EPI && p.row_scale != nullptr;                      // uniform over the launch
    constexpr bool ACC_FROM_BIAS_POSSIBLE = BIAS_INIT;

    // prefetch cursor: the K tile the next batch of DMAs fetches (pf_t of the output tile at (pf_m0, pf_n0)) and its stage;
    // bA / bW / nx are that K tile's byte offsets and LDS stage
    // A lane's byte offset of each of its sixteen DMA chunks lives in a VGPR of its own (vA / vW: the output tile's first row is
    // part of it — the descriptor's range check, which zero-fills the M / N tails, sees the VGPR offset only) and changes once
    // per output tile; the K tile's byte offset pf_k is the instruction's scalar offset: no vector arithmetic per DMA.
    int pf_v = v, pf_m0 = m0, pf_n0 = n0, pf_s = 0;
    unsigned pf_k = 0;
    int vA[8], vW[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int g = q & 3, row = (q >> 2) * 128 + wave * 32 + g * 8 + sub;            // row of the tile this lane fetches with DMA q
        const int keyW = PERM ? ((g << 1) | ((sub >> 1) & 1)) : sub;
        vA[q] = (int)((unsigned)(pf_m0 + row) * rbA) + (((lane & 7) ^ sub) << 4) - g * 1024;
        vW[q] = (int)((unsigned)(pf_n0 + row) * rbW) + (((lane & 7) ^ keyW) << 4) - g * 1024;
    }
    char* nx;
#define LW_PF_SETUP nx = smem + pf_s * LSTAGE;
// the cursor moves one K tile on: inside an output tile (NEXT), or from a tile's last K tile to the first of the workgroup's next
// tile (WRAP: the only place the tile order is evaluated — once per output tile, in a peeled K tile, so that the K loop carries
// no branch but its own back-edge: a taken branch stalls a lone wave for the length of its instruction refetch)
#define LW_PF_NEXT                                                            \
    pf_s ^= 1;                                                                \
    pf_k += LBK * 2;
#define LW_PF_WRAP                                                            \
    pf_s ^= 1;                                                                \
    pf_k = 0;                                                                 \
    pf_v += gridDim.x;                                                        \
    {                                                                         \
        int a_ = 0, b_ = 0;                                                   \
        /* past the last tile: a harmless prefetch nobody reads keeps the K tiles branch-free */ \
        if (pf_v < total) tile_of(pf_v, total, tiles_m, tiles_n, a_, b_, tgm); \
        const int dA_ = (int)((unsigned)(a_ * LBM - pf_m0) * rbA), dW_ = (int)((unsigned)(b_ * LBM - pf_n0) * rbW); \
        _Pragma("unroll") for (int q = 0; q < 8; ++q) { vA[q] += dA_; vW[q] += dW_; } \
        pf_m0 = a_ * LBM;                                                     \
        pf_n0 = b_ * LBM;                                                     \
    }
#define LW_DMA_ALL LW_DMA(0) LW_DMA(1) LW_DMA(2) LW_DMA(3) LW_DMA(4) LW_DMA(5) LW_DMA(6) LW_DMA(7) \
                   LW_DMA(8) LW_DMA(9) LW_DMA(10) LW_DMA(11) LW_DMA(12) LW_DMA(13) LW_DMA(14) LW_DMA(15)
    // K tiles 0 and 1 of the first output tile
    LW_PF_SETUP LW_DMA_ALL LW_PF_NEXT
    LW_PF_SETUP LW_DMA_ALL LW_PF_NEXT
    LW_PF_SETUP

    // this lane's fragment rows inside a stage (k-step 0 chunk; the k-step 1 chunk is the address ^ 64)
    const int wkey = PERM ? (((frow >> 2) << 1) | ((frow >> 1) & 1)) : (frow & 7);
    const unsigned a_row = (unsigned)(wm * LHALF + frow * 128 + ((fq ^ (frow & 7)) << 4));
    const unsigned b_row = (unsigned)((2 + wn2) * LHALF + (PERM ? ((frow >> 2) * 8 + (frow & 3)) : frow) * 128 + ((fq ^ wkey) << 4));
    const unsigned lds0 = (unsigned)(uintptr_t)smem;

    bf16x8 fa0[8], fb0[8], fa1[8], fb1[8];
    f32x4 acc[8][4];                      // one 128 x 64 strip, read out of the accumulator file for the epilogue
    u32x4 bias_cur[2][2] = {{u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}, {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}}};
    auto load_bias = [&](int n0_, u32x4 (&b)[2][2]) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
            for (int jq = 0; jq < 2; ++jq) {
                const int nb = n0_ + (wn2 * 2 + h) * 64 + jq * 32 + fq * 8;
                b[h][jq] = (p.bias && nb < p.N) ? *reinterpret_cast<const u32x4*>((const bf16_t*)p.bias + nb) : u32x4{0u, 0u, 0u, 0u};
            }
    };
    // column 32 (j >> 1) + 8 fq + 4 (j & 1) + r of strip h: element 4 (j & 1) + r of bias_cur[h][j >> 1]
    auto bias_of = [&](int h, int j, int r) -> float {
        const unsigned w = bias_cur[h][(j >> 1) & 1][((j & 1) * 4 + r) >> 1];
        return (r & 1) ? unpk_hi(w) : unpk_lo(w);
    };

#define PP_EPI_PRIV(E) ((E) + wave * 4096)
#ifndef PP_AUX_AHEAD     /* the row-dependent loads (residual / pos-embed) of ALL eight 16-row steps of a strip go out in front of its first store:
                            vmcnt retires in order, so a load issued behind a store is waited for behind that store's acknowledgement — a wave
                            that is alone on its SIMD sits through every one of those waits */
#define PP_AUX_AHEAD 7
#endif
#define PP_EPI_ATTR __attribute__((always_inline))
#define PP_EPI_STEP_HOOK(i)
#include "gemm_epilogue_wave.inc"
#undef PP_EPI_PRIV
#undef PP_EPI_ATTR
#undef PP_EPI_STEP_HOOK

    // SwiGLU: 8-byte stores straight from the fragments of one strip (gemm_pp.hip's form)
    auto epilogue_swiglu = [&]() __attribute__((always_inline)) {
        char* const Cw = (char*)p.C + (int64_t)(m0 + wm * 128) * p.ldc * 2;
        const unsigned ldc2 = (unsigned)p.ldc * 2u;
        const int m_lim = p.M - (m0 + wm * 128);
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int nin = n0 + wn * 64 + jj * 32;                  // wave-uniform
            if (nin < p.N) {
                const unsigned col2 = (unsigned)((nin >> 1) + fq * 4) * 2u;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float o[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) o[r] = silu_fast(acc[i][2 * jj][r]) * acc[i][2 * jj + 1][r];
                    if (i * 16 + frow < m_lim)
                        *reinterpret_cast<uint2*>(Cw + ((unsigned)(i * 16 + frow) * ldc2 + col2)) =
                            make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
                }
            }
        }
    };

    float rs[8];
    // one 128 x 64 strip h of the wave's two: accumulators -> registers -> (row scale, bias) -> the shared per-wave epilogue
    auto strip = [&](int h) __attribute__((always_inline)) {
        wn = wn2 * 2 + h;
        if (h == 0) lw_acc_copy<0>(acc, std::make_integer_sequence<int, 32>{});
        else lw_acc_copy<1>(acc, std::make_integer_sequence<int, 32>{});
        if (LW_DIAG_NOSTORE) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        if (RS_EPI && RS) {         // acc <- rstd[m] * acc + bias: the state the epilogue expects (the bias in the accumulators)
            float bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[j][r] = !BIAS_INIT ? 0.f : (h ? bias_of(1, j, r) : bias_of(0, j, r));
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(acc[i][j][r], rs[i], bv[j][r]);
        }
        if (EPI == GAR_EPI_SWIGLU) epilogue_swiglu();
        else epilogue_wave(smem + 2 * LSTAGE);
    };

    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");           // K tiles 0 and 1 of the first output tile (once per launch)
    __builtin_amdgcn_s_barrier();
    int sidx = 0;
#ifdef LW_TIMELINE
    unsigned tl_sum[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u}, tl_t = lw_now();
    const unsigned tl_begin = tl_t;
#endif
    while (true) {
        const int vn = v + gridDim.x;
        const bool has_next = vn < total;
        if (RS_EPI && RS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rs[i] = p.row_scale[min(m0 + wm * 128 + i * 16 + frow, p.M - 1)];
        }
        if (BIAS_INIT) load_bias(n0, bias_cur);
        const bool from_bias = ACC_FROM_BIAS_POSSIBLE && !RS;
        if (ACC_FROM_BIAS_POSSIBLE && from_bias) {
            float bv[8][4];
#pragma unroll
            for (int j = 0; j < 8; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[j][r] = bias_of(j >> 2, j & 3, r);
            lw_acc_fill(bv, std::make_integer_sequence<int, 64>{});
            asm volatile("s_nop 3" ::: "memory");
        }
        unsigned ra = (unsigned)(sidx * LSTAGE) + a_row, rb = (unsigned)(sidx * LSTAGE) + b_row;    // stage-relative
        // fragments of K tile 0, k-step 0 (exposed once per output tile)
        lw_rd<lw_woff<PERM, 0>::value>(fb0[0], lds0 + rb); lw_rd<lw_woff<PERM, 1>::value>(fb0[1], lds0 + rb);
        lw_rd<lw_woff<PERM, 2>::value>(fb0[2], lds0 + rb); lw_rd<lw_woff<PERM, 3>::value>(fb0[3], lds0 + rb);
        lw_rd<lw_woff<PERM, 4>::value>(fb0[4], lds0 + rb); lw_rd<lw_woff<PERM, 5>::value>(fb0[5], lds0 + rb);
        lw_rd<lw_woff<PERM, 6>::value>(fb0[6], lds0 + rb); lw_rd<lw_woff<PERM, 7>::value>(fb0[7], lds0 + rb);
        lw_rd<0 * 2048>(fa0[0], lds0 + ra); lw_rd<1 * 2048>(fa0[1], lds0 + ra); lw_rd<2 * 2048>(fa0[2], lds0 + ra); lw_rd<3 * 2048>(fa0[3], lds0 + ra);
        lw_rd<4 * 2048>(fa0[4], lds0 + ra); lw_rd<5 * 2048>(fa0[5], lds0 + ra); lw_rd<6 * 2048>(fa0[6], lds0 + ra); lw_rd<7 * 2048>(fa0[7], lds0 + ra);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        LW_TL(5)                  // tile top: next tile's coordinates, row scales / bias, the exposed fragment reads

        // One K tile: k-step 0 on F0 while F1 <- this stage's second k-chunk; [K tile g + 1 landed | barrier]; k-step 1 on F1 while
        // F0 <- the other stage's first k-chunk (K tile g + 1) and the DMAs of K tile g + 2 go into this stage.
#define LW_KTILE(Z, RD1, ADV)                                                                                     \
        LW_KSTEP(Z, true, false, fa0, fb0, fa1, fb1, lds0 + (ra ^ 64u), lds0 + (rb ^ 64u))                        \
        LW_TL(0)                                                                                                  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                          \
        LW_TL(1)                                                                                                  \
        asm volatile("s_barrier" ::: "memory");                                                                   \
        LW_TL(2)                                                                                                  \
        sidx ^= 1;                                                                                                \
        ra = (unsigned)(sidx * LSTAGE) + a_row;                                                                   \
        rb = (unsigned)(sidx * LSTAGE) + b_row;                                                                   \
        LW_KSTEP(false, RD1, true, fa1, fb1, fa0, fb0, lds0 + ra, lds0 + rb)                                      \
        LW_TL(3)                                                                                                  \
        ADV                                                                                                       \
        LW_PF_SETUP                                                                                               \
        LW_TL(4)
        // K tile 0 (cursor: K tile 2 of this output tile), K tiles 1 .. nt - 4 two per loop iteration, then the three K tiles
        // around the cursor's wrap into the next output tile: nt - 3 (fetches this tile's last K tile), nt - 2 and nt - 1 (the
        // next tile's K tiles 0 and 1; the last k-step reads no fragments: the epilogue follows)
        if (from_bias) { LW_KTILE(false, true, LW_PF_NEXT) } else { LW_KTILE(true, true, LW_PF_NEXT) }
        {
            int t = nt - 4;
#pragma nounroll
            for (; t >= 2; t -= 2) { LW_KTILE(false, true, LW_PF_NEXT) LW_KTILE(false, true, LW_PF_NEXT) }
            if (t) { LW_KTILE(false, true, LW_PF_NEXT) }
        }
        LW_KTILE(false, true, LW_PF_WRAP)
        LW_KTILE(false, true, LW_PF_NEXT)
        LW_KTILE(false, false, LW_PF_NEXT)
#undef LW_KTILE
        // epilogue of (m0, n0): the accumulators are complete 4 passes behind the last MFMA
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" ::: "memory");
#pragma nounroll
        for (int h = 0; h < 2; ++h) strip(h);
        LW_TL(6)                  // epilogue
        if (!has_next) break;
        v = vn;
        tile_of(v, total, tiles_m, tiles_n, tm, tn, tgm);
        m0 = tm * LBM;
        n0 = tn * LBM;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (unused) prefetches must land before the LDS is released
#ifdef LW_TIMELINE
    if (p.tokens_in == -777 && blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int q = 0; q < 7; ++q) ((unsigned*)p.pos)[wave * 8 + q] = tl_sum[q];
        ((unsigned*)p.pos)[wave * 8 + 7] = lw_now() - tl_begin;
    }
#endif
#endif
}

template <int EPI>
static void launch_lw(const gar_gemm_params& p, int pm, int pn, int num_cus, hipStream_t s) {
    constexpr int LDS = 2 * LSTAGE + LW_EPI_BYTES;
    static gar_once_per_device attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_lw_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    hipLaunchKernelGGL((gemm_bf16_lw_kernel<EPI>), dim3(min(pm * pn, num_cus)), dim3(256), LDS, s, p, pm, pn);   // persistent
}

// Which problems of the tile GEMM's domain (gar_gemm_pp_takes already holds) run in this frame: K >= LW_MIN_K — a lone wave per
// SIMD exposes its epilogue's latencies, which the longer main loops pay for — and the epilogues it instantiates.
#ifndef LW_MIN_K
#define LW_MIN_K 2048
#endif
bool gar_gemm_lw_try(const gar_gemm_params& p, hipStream_t s) {
#ifdef LW_OFF
    return false;
#endif
    if (p.K < LW_MIN_K || p.K < 4 * LBK) return false;       // (the K loop peels four K tiles)
    const int e_ = p.epilogue;
    const int num_cus = gar_num_cus();
    const int pm = (p.M + LBM - 1) / LBM, pn = (p.N + LBM - 1) / LBM;
    switch (e_) {
        case GAR_EPI_NONE: launch_lw<GAR_EPI_NONE>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_BIAS: launch_lw<GAR_EPI_BIAS>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_BIAS_SCALE_RES: launch_lw<GAR_EPI_BIAS_SCALE_RES>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_RES: launch_lw<GAR_EPI_RES>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_SWIGLU: launch_lw<GAR_EPI_SWIGLU>(p, pm, pn, num_cus, s); break;
        default: return false;
    }
    return true;
}
