// bf16 "ping-pong" GEMM: 256 x 256 x 64 tiles, 8 waves (2 M x 4 N, 128 x 64 per wave), ONE persistent workgroup per
// CU (128 KiB LDS), v_mfma_f32_16x16x32_bf16.
//
// Wave rows (wm = 0 / 1) share the four SIMDs of the CU and run the SAME instruction stream shifted by one barrier
// interval (row 1 executes one extra s_barrier up front): in every interval one row issues the 16 MFMAs of a phase
// while the other issues the ds_reads / global_load_lds of its next phase, so each SIMD's matrix pipe always has
// exactly one wave feeding it and LDS / VMEM issue hides behind MFMAs.
//
//   K tile = 4 phases (k-step, 64-row half of the wave's A rows): 8 / 4 / 8 / 4 ds_read_b128 and 16 MFMAs each (no fragment
//     is read twice; 32 VGPRs of fragments).
//   staging: the next K tile (of this output tile, or K tile 0 of the NEXT output tile of the persistent loop) is
//     DMA'd with `buffer_load_dwordx4 ... lds` (lane-linear LDS image, XOR swizzle applied to the SOURCE address,
//     M / N tails zero-filled by the descriptor's bounds check) into the other stage, one 1-KiB instruction per wave per
//     8-KiB unit, 1 + 4 + 2 + 1 instructions over the phases in the order the units are first read; vmcnt retires in order,
//     so `s_waitcnt vmcnt(n)` = "all but my newest n" waits for exactly the units the next barrier releases (see the
//     phase macros). The next output tile's first K tile is already in LDS when the epilogue starts.
//   loop shape: the K loop is rotated by one phase so that its back-edge (a taken branch costs ~150 cycles of
//     instruction refetch) sits in the short phase 3, and the next K tile's address arithmetic is issued between the MFMAs
//     of phase 3 (sched_group_barrier) — phase 0, 8 reads right behind the barrier that releases the stage, is the longest.
//   epilogue: the W rows of an n-tile pair are fed in a permuted order so that a lane ends up with EIGHT consecutive
//     output columns; every wave turns its own 128 x 64 strip into rows through a private 4-KiB piece of the
//     just-consumed LDS stage, 16 rows at a time, without workgroup barriers (epilogue_wave: bf16-staged when the output
//     is a function of accumulator and column, fp32-staged with row-side bias / LayerScale / residual / pos-embed / RoPE
//     otherwise); stores are 16-byte vectors, 8 rows x 128 contiguous bytes per instruction. QKV_ROPE with full sin / cos
//     tables keeps the older workgroup-level path (epilogue_lds: four 64-row fp32 chunks, 2 rows x 512 B per instruction);
//     SwiGLU stores straight from the fragments.
//   Measured (tools/gemm_timeline.py with -DPP_TIMELINE=n builds, tools/bench_gemm.py): main loop
//     1.45-1.59 PFLOP/s at a shader clock that the power limit holds at 1.4-1.65 GHz under this load (s_memtime ticks
//     per wall second), i.e. ~0.9 of the matrix pipes' rate at that clock; the workgroup-level epilogue cost ~21k clocks per
//     256 x 256 tile whatever the stores hit (HBM or one L2-resident tile), ~15k with hardware bf16 rounding; the per-wave
//     epilogue is worth another 2-11 % per GEMM on top.
#include <stdlib.h>
#include <type_traits>

#include "gemm_epilogue.h"

// Diagnostic builds only (tools/build_variant.sh, never in the product library): -DPP_NOSTORE runs the main loop without
// the epilogue, -DPP_L2STORE makes every tile store into the first tile's (L2-resident) region.
#ifdef PP_NOSTORE
#define PP_DIAG_NOSTORE true
#else
#define PP_DIAG_NOSTORE false
#endif
#ifdef PP_L2STORE
#define PP_DIAG_L2STORE true
#else
#define PP_DIAG_L2STORE false
#endif

#define PBM 256
#define PBK 64
#define PHALF (128 * 128)             // 16 KiB: 128 rows x 64 bf16
#define PSTAGE (4 * PHALF)            // [A rows 0-127 | A rows 128-255 | W rows 0-127 | W rows 128-255]

// swizzle keys (16-byte chunk index ^= key(row)), chosen per operand so its ds_read_b128 access sets are conflict-free:
//   A halves and SWIGLU W halves: fragment rows are 16 consecutive rows            -> key = row & 7
//   W halves otherwise: fragment rows are {8q + r + 4(j&1), q = 0..3, r = 0..3}     -> key = ((row>>3)&3)*2 + ((row>>1)&1)
// One 128-row half = 16 chunks of 8 rows (1 KiB, one wave-instruction each); wave w stages chunks w and 8+w.
// `buffer_load_dwordx4 ... lds` through a buffer descriptor: the per-lane part of the address is ONE 32-bit VGPR
// (voff = byte offset of (row = 8*wave + lane/8, 16-B chunk (lane&7)^key) inside a tile), everything else is scalar,
// and rows past the end of the tensor (M / N tails) are zero-filled by the descriptor's bounds check.
#ifndef PP_A_AUX          /* diagnostic builds: cache-policy bits of the A-operand DMA loads (2 = nt: stream through the L2) */
#define PP_A_AUX 0
#endif
#ifndef PP_W_AUX
#define PP_W_AUX 0
#endif
#ifndef PP_ST_AUX        /* diagnostic builds: cache-policy bits of the epilogue's output stores / residual loads (1 = sc0, 2 = nt, 16 = sc1) */
#define PP_ST_AUX 2          /* nt: +0.4 % weighted with the residual loads (profiles/r5_gemm_epilogue_traffic.txt 6) */
#endif
#ifndef PP_RES_AUX
#define PP_RES_AUX 2
#endif
template <int AUX = 0>
__device__ __forceinline__ void stage_half(__amdgpu_buffer_rsrc_t rsrc, int voff, unsigned row_bytes, unsigned base,
                                           char* lds, int wave) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_AS(lds + (i * 8 + wave) * 1024), 16,
                                                 voff + (int)(base + (unsigned)(i * 64) * row_bytes), 0, 0, AUX);
}

// one of the two instructions of stage_half: i = 0 -> rows 8*wave .. +7, i = 1 -> rows 64 + 8*wave .. +7 of the 128-row half
template <int AUX = 0>
__device__ __forceinline__ void dma1(__amdgpu_buffer_rsrc_t rsrc, int voff, unsigned row_bytes, unsigned base, char* lds,
                                     int wave, int i) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, LDS_AS(lds + (i * 8 + wave) * 1024), 16,
                                             voff + (int)(base + (unsigned)(i * 64) * row_bytes), 0, 0, AUX);
}

// 16-byte output stores of the pointer-addressed epilogues (q / k / v rows, patch-embed): streamed once, read by the
// NEXT kernel — non-temporal like the buffer stores of the linear epilogues (PP_NT_PTR_STORES=0: plain stores, diagnostic builds)
#ifndef PP_NT_PTR_STORES
#define PP_NT_PTR_STORES 1
#endif
__device__ __forceinline__ void pp_st8(bf16_t* p, const float (&v)[8]) {
    const u32x4 w = u32x4{pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7])};
    if (PP_NT_PTR_STORES) __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
    else *reinterpret_cast<u32x4*>(p) = w;
}

// GELU by table (BIAS_GELU epilogue): the erf GELU of a 256 x 256 tile costs ~46 VALU cycles per element as A&S 7.1.26
// (v_rcp + v_exp + 15 more) — 16k cycles per tile with the matrix pipes idle, 26 % of the fc1 GEMM (nostore / L2store /
// real-store builds). A 2048-interval table of (gelu(x_k), gelu(x_k+1) - gelu(x_k)) over [-8, 8) in the 16 KiB of
// LDS behind the two stages, filled with erff at kernel start, linearly interpolated: |error| <= h^2/8 max|gelu''| =
// (1/128)^2 / 8 * 0.8 = 6e-6 (bf16 rounds at 4e-3 relative), 9 VALU + one ds_read_b64 per element. x >= 8 returns x.
#define GELU_LUT_N 2048
#define GELU_LUT_BYTES (GELU_LUT_N * 8)
// Only the interval INDEX is clamped: beyond the table the last / first interval is extended linearly — gelu(x) = x to fp32
// precision for x >= 8 (the last interval's slope is 1) and 0 for x <= -8 (the first entry's slope is stored as 0) — so there
// is no range test. An entry holds the interval's LINE in table coordinates, (c0, s) with c0 = gelu(x_k) - s k, so that the
// value is one fma on u itself — neither floor(u) nor the weight u - floor(u) is formed: 5 VALU + one ds_read_b64 per element
// (fma, med3, cvt, shift, fma; 7 with the (value, slope) form, 9 with a range test). |c0| <= 18: its rounding adds <= 1e-6.
__device__ __forceinline__ float gelu_lut_u(float u, const char* lut) {          // u = 128 x + 1024: the table coordinate
    const int k = (int)__builtin_amdgcn_fmed3f(u, 0.0f, 2047.0f);                 // truncation = floor: the operand is >= 0
    const float2 e = *reinterpret_cast<const float2*>(lut + (k << 3));
    // one v_fma_f32 per element, pinned: left to itself the compiler pairs two elements into a v_pk_fma_f32 and pays three
    // v_mov to shuffle (line, slope) of the two table entries into operand pairs — 4 instructions where 2 do
    float r;
    asm("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(e.y), "v"(u), "v"(e.x));
    return r;
}
__device__ __forceinline__ float gelu_lut(float x, const char* lut) {
    return gelu_lut_u(__builtin_fmaf(x, 128.0f, 1024.0f), lut);
}

// Patch gather (GATHER, GAR_EPI_PATCH_POS only — gar_patch_embed): the A operand is not a matrix in HBM but the image
// tiles themselves. Row m = patch (tile, py, px) of a g x g = 32 x 32 patch grid; the K axis is re-ordered (and the
// weights with it, gar_patch_embed_weight_index) as  [tensor: pixel, mask][channel 0-2][q 0-3][ky = 4q + 0..3][16 kx slots],
// i.e. one K tile of 64 = four image rows of a patch x 16 pixel slots (patch <= 16 real ones, the rest meet zero weights).
// A lane's 16-byte DMA chunk is then 8 consecutive pixels of one image row — `buffer_load_dwordx4 ... lds` needs dword
// alignment only — 8 neighbouring patches are 8 * patch * 2 contiguous bytes of that row, 64 rows further down the tile
// are two patch rows further down the image, and a 256-row tile never leaves its image tile: the per-lane offset stays
// one VGPR and everything else scalar, exactly as for a dense A. The (c, ky) rows 14 / 15 and pixel slots 14 / 15 read
// the neighbouring patch / image row (finite data x zero weight; past the end of the tensor the descriptor returns 0).
struct gar_gather_args {
    const void* mask;      // binary mask, same [T, 3, img, img] layout as the pixels in p.A
    int img, patch;        // image side, patch side (g = img / patch == 32)
    unsigned bytes;        // T * 3 * img * img * 2
};
#define GATHER_KT_PER_TENSOR 12     // 3 channels x 4 K tiles (16 ky slots)

template <int EPI, bool GATHER = false>
__global__ __launch_bounds__(512, 2) void gemm_bf16_pp_kernel(const gar_gemm_params p, int tiles_m, int tiles_n,
                                                              const gar_gather_args ga) {
    constexpr bool PERM = EPI != GAR_EPI_SWIGLU;      // SwiGLU pairs (gate16 | up16) weight tiles and stores from the fragments
    constexpr bool LDS_EPI = true;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // 2 stages x 64 KiB (+ the GELU table)
    const int total = tiles_m * tiles_n;
    const int tid = threadIdx.x, lane = tid & 63;
    const char* const glut = smem + 2 * PSTAGE;
    if (EPI == GAR_EPI_BIAS_GELU) {       // published by the main loop's barriers long before the first epilogue reads it
#pragma unroll
        for (int j = 0; j < GELU_LUT_N / 512; ++j) {
            const int k = tid + 512 * j;
            const float x0 = -8.0f + (float)k * (1.0f / 128.0f);
            const float g0 = gelu_erf(x0), g1 = gelu_erf(x0 + 1.0f / 128.0f);
            // slopes are per table step; entry 0 extends to x < -8 with slope 0, entry 2047 to x > 8 with slope 1 / 128 per step;
            // stored as the line c0 + s u through (k, g0): c0 = g0 - s k, one rounding
            const float sl = k == 0 ? 0.0f : (k == GELU_LUT_N - 1 ? (x0 + 1.0f / 128.0f) - g0 : g1 - g0);
            *reinterpret_cast<float2*>(smem + 2 * PSTAGE + k * 8) = make_float2(__builtin_fmaf(-sl, (float)k, g0), sl);
        }
    }
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;
    const int frow = lane & 15, fq = lane >> 4;
    const int nt = p.K / PBK;

    f32x4 acc[8][4];
    bf16x8 af[4], bq[4];              // current A fragments (4 m-tiles of one half, one k-step), B fragments (4 n-tiles, one k-step)

    // row pitch in bytes (GATHER: 32 rows = one patch row of the image = patch * img * 2 bytes)
    const unsigned g_rb = GATHER ? (unsigned)ga.img * 2u : 0u;                  // image row
    const unsigned rbA = GATHER ? (unsigned)ga.patch * g_rb / 32u : (unsigned)p.lda * 2u, rbW = (unsigned)p.ldw * 2u;
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(
        (void*)A, 0, GATHER ? (int)ga.bytes : (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsA2 = GATHER ? __builtin_amdgcn_make_buffer_rsrc((void*)ga.mask, 0, (int)ga.bytes, 0x00020000) : rsA;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
    const int sub = lane >> 3;
    const int keyW = PERM ? (((wave & 3) << 1) | ((sub >> 1) & 1)) : sub;
    int voffA = (int)((unsigned)(wave * 8 + sub) * rbA) + (((lane & 7) ^ sub) << 4);
    if (GATHER) {       // row r = wave * 8 + sub of a 64-row block: patch (r >> 5, r & 31); chunk c8 = (image row c8 >> 1, half c8 & 1)
        const int r = wave * 8 + sub, c8 = (lane & 7) ^ sub;
        voffA = (int)((unsigned)(r >> 5) * (unsigned)ga.patch * g_rb + (unsigned)(r & 31) * (unsigned)ga.patch * 2u +
                      (unsigned)(c8 >> 1) * g_rb + (unsigned)(c8 & 1) * 16u);
    }
    // GATHER: byte offset of (first row m of a tile, K tile t) in its tensor, and which tensor
    auto g_base = [&](int m, int t) -> unsigned {
        const int ti = m >> 10, pr0 = (m & 1023) >> 5;
        const int tt = t >= GATHER_KT_PER_TENSOR ? t - GATHER_KT_PER_TENSOR : t;
        return ((unsigned)((ti * 3 + (tt >> 2)) * ga.img + pr0 * ga.patch + ((tt & 3) << 2))) * g_rb;
    };
    const int voffW = (int)((unsigned)(wave * 8 + sub) * rbW) + (((lane & 7) ^ keyW) << 4);
    auto stage_A = [&](int m0, int t, char* st) {
        const unsigned base = GATHER ? g_base(m0, t) : (unsigned)m0 * rbA + (unsigned)(t * PBK * 2);
        const __amdgpu_buffer_rsrc_t rs = GATHER && t >= GATHER_KT_PER_TENSOR ? rsA2 : rsA;
        stage_half<PP_A_AUX>(rs, voffA, rbA, base, st, wave);
        stage_half<PP_A_AUX>(rs, voffA, rbA, base + 128u * rbA, st + PHALF, wave);
    };
    auto stage_W = [&](int n0, int t, char* st) {
        const unsigned base = (unsigned)n0 * rbW + (unsigned)(t * PBK * 2);
        stage_half<PP_W_AUX>(rsW, voffW, rbW, base, st + 2 * PHALF, wave);
        stage_half<PP_W_AUX>(rsW, voffW, rbW, base + 128u * rbW, st + 3 * PHALF, wave);
    };

#ifdef PP_DEPHASE   /* diagnostic build (tools/runs/r2_measure3.sh): workgroups on odd XCDs (PP_DEPHASE = 1) or on odd CU slots of
                       every XCD (2) start half a tile period late, so that one half of the chip stores while the other
                       half computes */
    if ((PP_DEPHASE == 1 ? (blockIdx.x & 1) : ((blockIdx.x >> 3) & 1)) && total > (int)gridDim.x) {
        const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
        const unsigned long long d_ = (unsigned long long)p.K * 29ull;          // ~ half of (main loop + epilogue) of a tile
        while (__builtin_amdgcn_s_memtime() - t0_ < d_) __builtin_amdgcn_s_sleep(20);
    }
#endif
    int v = blockIdx.x, tm, tn;
    const int tgm = tile_group_m(p.K, tiles_n);
    tile_of(v, total, tiles_m, tiles_n, tm, tn, tgm);
    int m0 = tm * PBM, n0 = tn * PBM;
    // Bias epilogues (PERM layout): the accumulator chains start from the bias instead of 0 — the 128 adds per lane and tile,
    // and the epilogue's wait for its own bias load (s_waitcnt vmcnt(0) in front of the first store: in-order counter),
    // disappear. A lane's fragment columns are the same for all eight m-tiles: 16 values = two 16-byte loads per tile,
    // fetched one tile ahead. fp32 sum order changes from (products) + b to b + (products): one rounding of the bf16 result.
    constexpr bool BIAS_INIT = EPI == GAR_EPI_BIAS || EPI == GAR_EPI_BIAS_GELU || EPI == GAR_EPI_BIAS_SCALE_RES ||
                               EPI == GAR_EPI_QKV_ROPE;
    // epilogues that can consume a folded norm (row_scale) / produce the statistics of one (row_stats)
    constexpr bool RS_EPI = !GATHER && (EPI == GAR_EPI_NONE || EPI == GAR_EPI_BIAS || EPI == GAR_EPI_BIAS_GELU ||
                                        EPI == GAR_EPI_SWIGLU || EPI == GAR_EPI_QKV_ROPE ||
                                        EPI == GAR_EPI_QKV_ROPE_LLM);
    constexpr bool STATS_EPI = EPI == GAR_EPI_RES || EPI == GAR_EPI_BIAS_SCALE_RES;
    const bool RS = RS_EPI && p.row_scale != nullptr;                      // uniform over the launch
    u32x4 bias_cur[2] = {u32x4{0u, 0u, 0u, 0u}, u32x4{0u, 0u, 0u, 0u}};
    auto load_bias = [&](int n0_, u32x4 (&b)[2]) {
#pragma unroll
        for (int jq = 0; jq < 2; ++jq) {
            const int nb = n0_ + wn * 64 + jq * 32 + fq * 8;
            b[jq] = (p.bias && nb < p.N) ? *reinterpret_cast<const u32x4*>((const bf16_t*)p.bias + nb) : u32x4{0u, 0u, 0u, 0u};
        }
    };
    if (BIAS_INIT) load_bias(n0, bias_cur);
    stage_A(m0, 0, smem);
    stage_W(n0, 0, smem);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();      // shift wave row 1 by one interval

    // byte offsets of this lane's fragment rows inside a stage
    const int a_row = wm * PHALF + frow * 128;                                   // + (mh*4+i)*2048
    const int ca0 = ((0 * 4 + fq) ^ (lane & 7)) << 4, ca1 = ((1 * 4 + fq) ^ (lane & 7)) << 4;
    // W fragment row of n-tile j for this lane (local to the wave's 64-row strip)
    //   PERM:   32*(j>>1) + (frow>>2)*8 + (frow&3) + 4*(j&1)      key = (frow>>2)*2 + ((frow>>1)&1)  (same for all j)
    //   SWIGLU: 16*j + frow                                          key = frow & 7
    const int wkey = PERM ? (((frow >> 2) << 1) | ((frow >> 1) & 1)) : (frow & 7);
    const int b_row = (2 + (wn >> 1)) * PHALF +
                      ((wn & 1) * 64 + (PERM ? ((frow >> 2) * 8 + (frow & 3)) : frow)) * 128;
    const int cb0 = ((0 * 4 + fq) ^ wkey) << 4, cb1 = ((1 * 4 + fq) ^ wkey) << 4;
    constexpr int BJ0 = PERM ? 4 * 128 : 16 * 128;      // byte step from n-tile 2q to 2q+1
    constexpr int BJ1 = 32 * 128;                       // byte step from n-tile pair q to q+1

#define READ_A(st, mh, kk)                                                                               \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                         \
        af[i] = *reinterpret_cast<const bf16x8*>(smem + ((st) ^ ((kk) ? 64 : 0)) + ((mh) * 4 + i) * 2048);
#define READ_B(st, kk)                                                                                   \
    _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                         \
        bq[j] = *reinterpret_cast<const bf16x8*>(smem + ((st) ^ ((kk) ? 64 : 0)) + (j >> 1) * BJ1 + (j & 1) * BJ0);
// Issue order of a phase's 16 MFMAs (4 m-tiles x 4 n-tiles, independent accumulators): n-tile outer, m-tiles in SNAKE order, so
// that consecutive MFMAs always share one operand fragment — the W fragment for four in a row, the A fragment across the turn.
// The chip is power-limited in this loop (profiles/r6_gemm_lone_wave.txt), and an operand that does not change between two MFMAs
// does not toggle its lanes: +0.9 % weighted over the planner's shapes against the row-major order (every fourth MFMA changed BOTH
// operands), same box, 4 x 8 repetitions alternated (profiles/r6_gemm_mfma_order.txt); outputs bit-identical (order of issue only).
#ifndef PP_MMA_ORDER      /* diagnostic builds: 0 = m-tile outer, n-tiles 0..3 (rounds 2 - 5), 1 = that in snake order, 2 = n-tile outer, 3 = n-tile outer, snake */
#define PP_MMA_ORDER 3
#endif
#define MMA(mh)                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                       \
    if (PP_MMA_ORDER >= 2) {                                                                             \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                     \
            _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                            \
                const int i = (PP_MMA_ORDER == 3 && (j & 1)) ? 3 - ii : ii;                              \
                acc[(mh) * 4 + i][j] = MFMA_16x16x32(bq[j], af[i], acc[(mh) * 4 + i][j]);                \
            }                                                                                            \
    } else {                                                                                             \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                     \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                            \
                const int j = (PP_MMA_ORDER == 1 && (i & 1)) ? 3 - jj : jj;                              \
                acc[(mh) * 4 + i][j] = MFMA_16x16x32(bq[j], af[i], acc[(mh) * 4 + i][j]);                \
            }                                                                                            \
    }                                                                                                    \
    __builtin_amdgcn_s_setprio(0);
#ifdef PP_TIMELINE   /* diagnostic build (tools/gemm_timeline.py): shader-clock stamps around every barrier */
#if PP_TIMELINE >= 2   /* light: only the stamps around phase 0 (barriers 7 -> 0), so the other phases run undisturbed */
#define TL(i) if ((i) == 15 || (i) <= 1 || (PP_TIMELINE == 3 && (i) >= 13)) tl[i] = (unsigned)__builtin_amdgcn_s_memtime();
#define TLX(i) __builtin_amdgcn_sched_barrier(0); tl[i] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0);
#else
#define TL(i) tl[i] = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#else
#define TL(i)
#endif
#if PP_TIMELINE == 4     /* K-tile durations by position inside an output tile (0, 1, 2, 3, later) + epilogue */
#define TL_ACC4(kt)                                                                              \
    if ((kt) >= 1) {                                                                             \
        const unsigned d_ = tl[15] - tl13p;                                                      \
        const int b_ = (kt) - 1 < 4 ? (kt) - 1 : 4;                                              \
        tl_sum[8] += b_ == 0 ? d_ : 0; tl_sum[9] += b_ == 1 ? d_ : 0; tl_sum[10] += b_ == 2 ? d_ : 0; \
        tl_sum[11] += b_ == 3 ? d_ : 0; tl_sum[12] += b_ == 4 ? d_ : 0;                           \
    }
#else
#define TL_ACC4(kt)
#endif
#if PP_TIMELINE == 3
#define TLX3(i) TLX(i)
#else
#define TLX3(i)
#endif
#if PP_TIMELINE >= 2
#define TL_ACCUMULATE                                                                            \
    tl_sum[0] += tl[0] - tl[15];              /* phase-0 loads */                                \
    tl_sum[1] += tl[1] - tl[0];               /* wait at barrier 0 */                            \
    tl_sum[15] += tl[15] - tl13p;             /* whole K tile (release of barrier 7 to the next) */ \
    tl13p = tl[15];                                                                              \
    if (PP_TIMELINE == 3) {                   /* previous K tile's phase-3 MFMA part (stamps lag one K tile) */ \
        tl_sum[2] += tl2p - tl13q;            /* barrier 6 release -> 16 MFMAs issued */           \
        tl_sum[3] += tl[3] - tl2p;            /* next K tile's address preparation */              \
        tl_sum[4] += tl[14] - tl[3];          /* vmcnt wait */                                     \
        tl_sum[5] += tl[15] - tl[14];         /* wait at barrier 7 */                              \
        tl2p = tl[2]; tl13q = tl[13];                                                             \
    }
#elif defined(PP_TIMELINE)
#define TL_ACCUMULATE                                                                            \
    _Pragma("unroll") for (int q = 0; q < 7; ++q) {                                               \
        tl_sum[2 * q] += tl[2 * q] - (q ? tl[2 * q - 1] : tl[15]);       /* work before barrier q */ \
        tl_sum[2 * q + 1] += tl[2 * q + 1] - tl[2 * q];                  /* wait at barrier q */    \
    }                                                                                            \
    tl_sum[14] += tl[14] - tl13p;                                        /* barrier 7 of the previous K tile */ \
    tl_sum[15] += tl[15] - tl[14];                                                               \
    tl13p = tl[13];
#else
#define TL_ACCUMULATE
#endif
#define VMWAIT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
#define BAR_THEN_WAIT_LDS(i)                                 \
    __builtin_amdgcn_sched_barrier(0);                       \
    TL(2 * (i))                                              \
    __builtin_amdgcn_s_barrier();                            \
    TL(2 * (i) + 1)                                          \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       \
    __builtin_amdgcn_sched_barrier(0);
#define BAR(i)                                               \
    __builtin_amdgcn_sched_barrier(0);                       \
    TL(2 * (i))                                              \
    __builtin_amdgcn_s_barrier();                            \
    TL(2 * (i) + 1)                                          \
    __builtin_amdgcn_sched_barrier(0);

#ifdef PP_TIMELINE
    unsigned tl[16], tl_sum[16], tl13p, tl2p = 0, tl13q = 0, tl_e = 0;
    const unsigned long long tl_start = __builtin_amdgcn_s_memtime();
#pragma unroll
    for (int q = 0; q < 16; ++q) tl_sum[q] = 0;
    tl[2] = tl[3] = tl[13] = tl[14] = tl[15] = tl13p = tl2p = tl13q = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    auto epilogue = [&]() {
        if (PP_DIAG_NOSTORE) {              // diagnostic build (-DPP_NOSTORE, tools/build_variant.sh): skip the epilogue, keep acc live
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(acc[i][j]));
            return;
        }
        if (EPI == GAR_EPI_SWIGLU) {
            // 8-byte stores straight from the fragments (gate16 | up16 tile pairs: lane (frow, fq) holds 4 consecutive output
            // columns of row frow). Per-tile base + one v_mad_u32_u24 per store; the tile GEMM's entry conditions (N % 8 == 0,
            // ldc % 8 == 0, 16-byte aligned C) make every store aligned and whole, so the generic store's checks are gone.
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
                            // (plain stores: as non-temporal 8-byte stores — half the lines written in two pieces — this GEMM is 11 % slower)
                            *reinterpret_cast<uint2*>(Cw + ((unsigned)(i * 16 + frow) * ldc2 + col2)) =
                                make_uint2(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]));
                    }
                }
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int m = m0 + wm * 128 + i * 16 + frow;
            if (m < p.M) {
                if (EPI == GAR_EPI_SWIGLU) {
                } else {
#pragma unroll
                    for (int jq = 0; jq < 2; ++jq) {
                        const int n = n0 + wn * 64 + jq * 32 + fq * 8;
                        if (n < p.N) {
                            float o[8] = {acc[i][2 * jq][0], acc[i][2 * jq][1], acc[i][2 * jq][2], acc[i][2 * jq][3],
                                          acc[i][2 * jq + 1][0], acc[i][2 * jq + 1][1], acc[i][2 * jq + 1][2],
                                          acc[i][2 * jq + 1][3]};
                            epilogue_store8<bf16_t, EPI>(p, m, n, o);
                        }
                    }
                }
            }
        }
    };

    // Row-coalesced epilogue through LDS (PERM epilogues): the just-consumed stage (64 KiB) holds 64 output rows x 256
    // columns of fp32 accumulators at a time; the owning wave row writes its fragments (ds_write_b128, XOR-swizzled
    // 16-byte chunks), then ALL eight waves read whole rows back and run bias / LayerScale / residual / store on
    // 8-column groups: every global access of a wave instruction is 2 rows x 512 contiguous bytes.
    // Called with both wave rows aligned; 2 barriers per 64-row chunk.
#if PP_TIMELINE == 5   /* epilogue: per 64-row chunk [aux loads + LDS write | barrier | LDS read + math + stores | barrier] */
#define EPI_TL(i) { const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); tl_sum[i] += t_ - tl_e; tl_e = t_; }
#else
#define EPI_TL(i)
#endif
    auto epilogue_lds = [&](char* E) {
        constexpr bool QKV = EPI == GAR_EPI_QKV_ROPE;
        constexpr bool HAS_RES = EPI == GAR_EPI_BIAS_SCALE_RES || EPI == GAR_EPI_RES;
        // a thread always serves the same 8-column group (cg = tid & 31): bias / LayerScale are loaded once per tile
        // (opaque per call: otherwise the compiler hoists the sixteen 64-bit row addresses and the LDS offsets out of the
        // persistent tile loop, spills them, and reloads each one behind an s_waitcnt vmcnt(0) — i.e. behind the previous store)
        int cg = tid & 31, r0 = tid >> 5, frow_e = frow, fq_e = fq;
        asm volatile("" : "+v"(cg), "+v"(r0), "+v"(frow_e), "+v"(fq_e));
        const int n = n0 + cg * 8;
        const bool nok = n < p.N;
        float gam8[8];
        // QKV_ROPE: this thread's 8 columns are dims d..d+7 of head h of q (part 0), k (1) or v (2)
        const int Da = p.qkv_heads * p.qkv_head_dim;
        const int part = QKV ? n / Da : 0;
        const int nn = QKV ? n - part * Da : 0;
        const int qh = QKV ? nn / p.qkv_head_dim : 0, qd = QKV ? nn - qh * p.qkv_head_dim : 0;
        if (EPI == GAR_EPI_BIAS_SCALE_RES && nok) ld8((const bf16_t*)p.gamma + n, gam8);
        auto write_chunk = [&](int c) {        // the owning wave row's fragments of 64-row chunk c -> E (fp32)
            if (wm == (c >> 1)) {
#pragma unroll
                for (int il = 0; il < 4; ++il) {
                    const int i = (c & 1) * 4 + il;
                    const int row = il * 16 + frow_e;
                    char* rp = E + row * 1024;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int chunk = wn * 16 + (j >> 1) * 8 + fq_e * 2 + (j & 1);
                        *reinterpret_cast<f32x4*>(rp + ((chunk ^ (frow_e & 15)) << 4)) = acc[i][j];
                    }
                }
            }
        };
        // Pipelined over the four chunks: [read chunk c back into registers] barrier [write chunk c+1 | math + stores of
        // chunk c] barrier. The stores of a chunk are issued against the chip-wide write rate (all CUs reach their
        // epilogues together), so the LDS hand-over of the next chunk runs under that back-pressure instead of after it.
        write_chunk(0);
        EPI_TL(0) __builtin_amdgcn_s_barrier(); EPI_TL(1)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            // issue this chunk's row-dependent loads (residual / pos-embed) before waiting for the LDS hand-over
            uint4 aux[4];
            int64_t off[4];
            bool ok[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int m = m0 + c * 64 + k * 16 + r0;
                ok[k] = nok && m < p.M;
                if (EPI == GAR_EPI_PATCH_POS) {
                    const int tile = m / p.tokens_in;
                    const int tok = p.token_offset + (m - tile * p.tokens_in);
                    off[k] = ((int64_t)tile * p.tokens_out + tok) * p.ldc + n;
                    if (ok[k]) aux[k] = *reinterpret_cast<const uint4*>((const bf16_t*)p.pos + (int64_t)tok * p.N + n);
                } else {
                    off[k] = (int64_t)m * p.ldc + n;
                    if (PP_DIAG_L2STORE) off[k] = (int64_t)(m - m0) * p.ldc + (n - n0);   // diagnostic build: L2-resident stores
                    if (HAS_RES && ok[k])
                        aux[k] = *reinterpret_cast<const uint4*>((const bf16_t*)p.residual + (int64_t)m * p.ldr + n);
                }
            }
            f32x4 ra4[4], rb4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = k * 16 + r0;
                const char* rp = E + row * 1024;
                ra4[k] = *reinterpret_cast<const f32x4*>(rp + (((cg * 2) ^ (row & 15)) << 4));
                rb4[k] = *reinterpret_cast<const f32x4*>(rp + (((cg * 2 + 1) ^ (row & 15)) << 4));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            EPI_TL(2) __builtin_amdgcn_s_barrier(); EPI_TL(3)
            if (c < 3) write_chunk(c + 1);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const f32x4 a = ra4[k], b = rb4[k];
                if (ok[k]) {
                    float o[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
                    if (EPI == GAR_EPI_BIAS_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = gelu_fast(o[e]);
                    }
                    if (HAS_RES || EPI == GAR_EPI_PATCH_POS) {
                        const unsigned int w[4] = {aux[k].x, aux[k].y, aux[k].z, aux[k].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float lo = unpk_lo(w[e]), hi = unpk_hi(w[e]);
                            if (EPI == GAR_EPI_BIAS_SCALE_RES) {
                                o[2 * e] = lo + gam8[2 * e] * o[2 * e];
                                o[2 * e + 1] = hi + gam8[2 * e + 1] * o[2 * e + 1];
                            } else {
                                o[2 * e] += lo;
                                o[2 * e + 1] += hi;
                            }
                        }
                    }
                    if (QKV) {
                        const int m = m0 + c * 64 + k * 16 + r0;
                        const int tile = m / p.qkv_tokens, tok = m - tile * p.qkv_tokens;
                        if (part < 2) {
                            if (tok >= p.qkv_prefix) {      // rot(x) = (-x[2i+1], x[2i]) on interleaved pairs
                                float s8[8], c8[8];
                                const int64_t ro = (int64_t)(tok - p.qkv_prefix) * p.qkv_head_dim + qd;
                                ld8(p.qkv_sin + ro, s8);
                                ld8(p.qkv_cos + ro, c8);
#pragma unroll
                                for (int e = 0; e < 8; e += 2) {
                                    const float x0 = o[e], x1 = o[e + 1];
                                    o[e] = x0 * c8[e] + (-x1) * s8[e];
                                    o[e + 1] = x1 * c8[e + 1] + x0 * s8[e + 1];
                                }
                            }
                            if (part == 0) {
#pragma unroll
                                for (int e = 0; e < 8; ++e) o[e] *= p.qkv_q_scale;
                            }
                            bf16_t* dst = (bf16_t*)(part == 0 ? p.qkv_q : p.qkv_k) +
                                          (((int64_t)tile * p.qkv_heads + qh) * p.qkv_tokens_pad + tok) * p.qkv_head_dim + qd;
                            st8(dst, o);
                        } else if (p.qkv_v) {        // v head-major like k (consumed by gar_attention_vrow: no transpose pass)
                            st8((bf16_t*)p.qkv_v +
                                    (((int64_t)tile * p.qkv_heads + qh) * p.qkv_tokens_pad + tok) * p.qkv_head_dim + qd, o);
                        } else {
                            st8((bf16_t*)p.C + (int64_t)m * p.ldc + nn, o);
                        }
                    } else {
                        st8((bf16_t*)p.C + off[k], o);
                    }
                }
            }
            if (c < 3) { EPI_TL(0) __builtin_amdgcn_s_barrier(); EPI_TL(1) }
        }
    };

#define PP_EPI_PRIV(E) ((E) + wm * PHALF + wn * 4096)
#define PP_EPI_ATTR
#define PP_EPI_STEP_HOOK(i) if ((i) == 0 && wm == 0) __builtin_amdgcn_s_barrier();      /* un-stagger: row 1 has finished its last MFMAs by now */
#include "gemm_epilogue_wave.inc"

    int sidx = 0;
    while (true) {
        const int vn = v + gridDim.x;
        const bool has_next = vn < total;
        int m0n = 0, n0n = 0;
        if (has_next) {
            int tmn, tnn;
            tile_of(vn, total, tiles_m, tiles_n, tmn, tnn, tgm);
            m0n = tmn * PBM;
            n0n = tnn * PBM;
        }
        // Folded norm, consumer side (gar_gemm_params.row_scale): the accumulator rows are multiplied by rstd[m] before the
        // epilogue and the bias is added AFTER that scale — so the chains start from 0 and bias_cur holds THIS tile's bias.
        float rs[8];
        if (RS_EPI && RS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) rs[i] = p.row_scale[min(m0 + wm * 128 + i * 16 + frow, p.M - 1)];
        }
        if (BIAS_INIT && RS) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            load_bias(n0, bias_cur);
        } else if (BIAS_INIT) {
            // acc[i][j][r] is column 32 (j >> 1) + 8 fq + 4 (j & 1) + r of the wave's strip: element 4 (j & 1) + r of bias_cur[j >> 1]
            float bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned w = bias_cur[j >> 1][((j & 1) * 4 + r) >> 1];
                    bv[j][r] = (r & 1) ? unpk_hi(w) : unpk_lo(w);
                }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{bv[j][0], bv[j][1], bv[j][2], bv[j][3]};
            // the NEXT tile's bias: issued here, older than every DMA of this tile, so the main loop's in-order vmcnt(n)
            // waits cover it long before it is used at the top of the next iteration
            load_bias(n0n, bias_cur);
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // loop-carried, prepared at the end of the previous K tile under its last MFMAs (the first interval of a K tile is
        // the longest: nothing but the reads and one DMA should sit between the barrier and the next barrier)
        int ra = sidx * PSTAGE + a_row + ca0;                // this lane's A / W fragment rows (k-step 0 chunk) in the stage being read
        int rb = sidx * PSTAGE + b_row + cb0;
        char* nx = smem + (sidx ^ 1) * PSTAGE;               // stage being filled
        unsigned bA, bW;                                     // global byte offsets of the K tile being prefetched
        __amdgpu_buffer_rsrc_t rsAc = rsA;                   // GATHER: the tensor (pixels / mask) that K tile comes from
        {
            const bool last = nt == 1;
            const int pm = last ? m0n : m0, pn = last ? n0n : n0, pt = last ? 0 : 1;
            bA = GATHER ? g_base(pm, pt) : (unsigned)pm * rbA + (unsigned)(pt * PBK * 2);
            bW = (unsigned)pn * rbW + (unsigned)(pt * PBK * 2);
            if (GATHER) rsAc = pt >= GATHER_KT_PER_TENSOR ? rsA2 : rsA;
        }
        // Four phases per K tile = (k-step, 64-row half of the wave's A rows): 8 + 4 + 8 + 4 fragment reads.
        // The eight 1-KiB DMAs a wave issues per K tile go out 1 + 4 + 2 + 1 over the phases, ordered by when their 8-KiB
        // unit (one instruction from each of the 8 waves) is first read in the next K tile:
        //   W half 0 / 1, rows 0-63 / 64-127 (a b c d) and A half 0 rows 0-63 (e): row 0's phase 0
        //   A half 1 rows 0-63 (f): row 1's phase 0 — one interval later;  A half 0 rows 64-127 (g): row 0's phase 1;
        //   A half 1 rows 64-127 (h): row 1's phase 1.
        // vmcnt retires in order, so "all but my newest n" names exactly the units that must have landed before the barrier
        // ahead of each first read; the same counts are right for both rows (row 1 runs the same program one interval
        // later, so the count its deadline needs is never looser than row 0's at the same program point).
        // After the very last tile m0n = n0n = 0: a harmless prefetch nobody reads keeps the phases branch-free.
#define DMA_A dma1<PP_W_AUX>(rsW, voffW, rbW, bW, nx + 2 * PHALF, wave, 0);
#define DMA_B dma1<PP_W_AUX>(rsW, voffW, rbW, bW, nx + 2 * PHALF, wave, 1);
#define DMA_C dma1<PP_W_AUX>(rsW, voffW, rbW, bW + 128u * rbW, nx + 3 * PHALF, wave, 0);
#define DMA_D dma1<PP_W_AUX>(rsW, voffW, rbW, bW + 128u * rbW, nx + 3 * PHALF, wave, 1);
#define DMA_E dma1<PP_A_AUX>(rsAc, voffA, rbA, bA, nx, wave, 0);
#define DMA_F dma1<PP_A_AUX>(rsAc, voffA, rbA, bA + 128u * rbA, nx + PHALF, wave, 0);
#define DMA_G dma1<PP_A_AUX>(rsAc, voffA, rbA, bA, nx, wave, 1);
#define DMA_H dma1<PP_A_AUX>(rsAc, voffA, rbA, bA + 128u * rbA, nx + PHALF, wave, 1);
// 1 + 4 + 2 + 1 (1 5 2 0 and 0 5 2 1 measured within noise of it)
#define L0_DMA DMA_A
#define L1_DMA DMA_B DMA_C DMA_D DMA_E
#define L2_DMA DMA_F DMA_G
#define L3_DMA DMA_H
#define L0_WAIT VMWAIT(2)
#define M0_WAIT VMWAIT(1)
#define PHASE012                                                                                                     \
        /* ---- phase 0: k-step 0, rows 0-63 */                                                                      \
        READ_A(ra, 0, 0)                                                                                             \
        READ_B(rb, 0)                                                                                                \
        L0_DMA                                                                                                       \
        L0_WAIT                                                                /* g landed (row 0's phase-1 reads) */ \
        BAR_THEN_WAIT_LDS(0)                                                                                         \
        MMA(0)                                                                                                       \
        M0_WAIT                                                                /* h landed (row 1's phase-1 reads) */ \
        BAR(1)                                                                                                       \
        /* ---- phase 1: k-step 0, rows 64-127 */                                                                    \
        READ_A(ra, 1, 0)                                                                                             \
        L1_DMA                                                                                                       \
        BAR_THEN_WAIT_LDS(2)                                                                                         \
        MMA(1)                                                                                                       \
        BAR(3)                                                                                                       \
        /* ---- phase 2: k-step 1, rows 0-63 */                                                                      \
        READ_A(ra, 0, 1)                                                                                             \
        READ_B(rb, 1)                                                                                                \
        L2_DMA                                                                                                       \
        BAR_THEN_WAIT_LDS(4)                                                                                         \
        MMA(0)                                                                                                       \
        BAR(5)
        // phase 3 of K tile kt; under its MFMAs the addresses of K tile kt + 1 (whose prefetch is K tile kt + 2, wrapping into
        // the next output tile) are prepared, so that nothing but reads and one DMA sits in the next phase 0
#define PHASE3(kt)                                                                                                   \
        /* ---- phase 3: k-step 1, rows 64-127 */                                                                    \
        READ_A(ra, 1, 1)                                                                                             \
        L3_DMA                                                                                                       \
        VMWAIT(3)                                                              /* a-e landed (row 0's next phase 0) */ \
        BAR_THEN_WAIT_LDS(6)                                                                                         \
        /* the next K tile's addresses, issued between this phase's MFMAs (two scalar / vector ALU slots behind each) so  \
           that nothing but reads and DMAs sits in the next phase 0; pinned so they are not re-derived behind the barrier */ \
        __builtin_amdgcn_s_setprio(1);                                                                               \
        sidx ^= 1;                                                                                                   \
        ra = sidx * PSTAGE + a_row + ca0;                                                                            \
        rb = sidx * PSTAGE + b_row + cb0;                                                                            \
        int nxo_ = (sidx ^ 1) * PSTAGE;                                                                              \
        {                                                                                                            \
            const bool last_ = (kt) + 2 >= nt;                                                                       \
            const int pm_ = last_ ? m0n : m0, pn_ = last_ ? n0n : n0, pt_ = last_ ? 0 : (kt) + 2;                    \
            bA = GATHER ? g_base(pm_, pt_) : (unsigned)pm_ * rbA + (unsigned)(pt_ * PBK * 2);                       \
            bW = (unsigned)pn_ * rbW + (unsigned)(pt_ * PBK * 2);                                                    \
            if (GATHER) rsAc = pt_ >= GATHER_KT_PER_TENSOR ? rsA2 : rsA;                                             \
        }                                                                                                            \
        if (PP_MMA_ORDER >= 2) {                                                                                     \
            _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                             \
                _Pragma("unroll") for (int ii = 0; ii < 4; ++ii) {                                                    \
                    const int i = (PP_MMA_ORDER == 3 && (j & 1)) ? 3 - ii : ii;                                      \
                    acc[4 + i][j] = MFMA_16x16x32(bq[j], af[i], acc[4 + i][j]);                                      \
                }                                                                                                    \
        } else {                                                                                                     \
            _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                             \
                _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                    \
                    const int j = (PP_MMA_ORDER == 1 && (i & 1)) ? 3 - jj : jj;                                      \
                    acc[4 + i][j] = MFMA_16x16x32(bq[j], af[i], acc[4 + i][j]);                                      \
                }                                                                                                    \
        }                                                                                                            \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                       \
            __builtin_amdgcn_sched_group_barrier(0x006, 2, 0);                                                       \
        }                                                                                                            \
        __builtin_amdgcn_s_setprio(0);                                                                               \
        asm volatile("" : "+v"(ra), "+v"(rb), "+s"(nxo_), "+s"(bA), "+s"(bW));                                       \
        nx = smem + nxo_;                                                                                            \
        TLX3(2)                                                                                                      \
        TL_ACC4(kt)                                                                                                  \
        TL_ACCUMULATE                                                                                                \
        TLX3(3)                                                                                                      \
        VMWAIT(2)                                                              /* f landed (row 1's next phase 0) */ \
        BAR(7)
        // The K loop is rotated by one phase: a taken branch costs the wave ~150 cycles of instruction refetch (measured:
        // tools/hw_probes.hip, tools/gemm_timeline.py), and phase 3's load interval (4 reads, 1 DMA) is the one with that
        // much slack — phase 0's (8 reads behind the barrier that releases the stage) is the longest.
#ifdef PP_STOREOVERLAP   /* diagnostic build: the first K tile of an output tile is completely in LDS before the previous
                            tile's epilogue starts (vmcnt(0) below), so its two waits are dropped — they would wait for the
                            epilogue's STORES as well (vmcnt retires in order and counts stores) */
#undef L0_WAIT
#undef M0_WAIT
#define L0_WAIT
#define M0_WAIT
        PHASE012
#undef L0_WAIT
#undef M0_WAIT
#define L0_WAIT VMWAIT(2)
#define M0_WAIT VMWAIT(1)
#else
        PHASE012
#endif
#pragma nounroll
        for (int t = 1; t < nt; ++t) {
            PHASE3(t - 1)
            PHASE012
        }
        PHASE3(nt - 1)
#ifdef PP_STOREOVERLAP
        VMWAIT(0)
#endif
#undef PHASE012
#undef PHASE3
        // un-stagger (row 0 waits one interval for row 1), run the epilogue of (m0, n0) on both rows at the same time
        // — the next tile's first K tile is already in LDS, the stores drain under its main loop — then re-stagger.
#if PP_TIMELINE == 5
        tl_e = (unsigned)__builtin_amdgcn_s_memtime();
#endif
#if PP_TIMELINE == 4
        const unsigned tl_e0 = (unsigned)__builtin_amdgcn_s_memtime();
        tl_sum[12] += tl_e0 - tl[15];       // (the last K tile's barrier-7 release to here: ~0, keeps tl[15] live)
#endif
        if (RS_EPI && RS) {
            // acc <- rstd[m] * acc + bias: the state every epilogue below expects at its entry (the bias in the accumulators)
            float bv[4][4];
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const unsigned w = bias_cur[j >> 1][((j & 1) * 4 + r) >> 1];
                    bv[j][r] = !BIAS_INIT ? 0.f : ((r & 1) ? unpk_hi(w) : unpk_lo(w));
                }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[i][j][r] = __builtin_fmaf(acc[i][j][r], rs[i], bv[j][r]);
        }
        constexpr bool WAVE_EPI = EPI != GAR_EPI_SWIGLU;
        if (PERM && LDS_EPI && WAVE_EPI && !PP_DIAG_NOSTORE && (EPI != GAR_EPI_QKV_ROPE || (p.qkv_cos == nullptr && p.qkv_v != nullptr))) {
            epilogue_wave(smem + (sidx ^ 1) * PSTAGE);      // starts at once in each wave row; un-staggers inside
        } else {
            if (wm == 0) __builtin_amdgcn_s_barrier();
            if (PERM && LDS_EPI && !PP_DIAG_NOSTORE) epilogue_lds(smem + (sidx ^ 1) * PSTAGE);
            else epilogue();
        }
#if PP_TIMELINE == 4
        tl_sum[13] += (unsigned)__builtin_amdgcn_s_memtime() - tl_e0;          // un-stagger + epilogue
#endif
#ifdef PP_TIMELINE
        tl[2] = tl[3] = tl[13] = tl[14] = tl[15] = tl13p = tl2p = tl13q = (unsigned)__builtin_amdgcn_s_memtime();
#endif
        if (!has_next) break;
        v = vn;
        m0 = m0n;
        n0 = n0n;
        if (wm == 1) __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the last (unused) prefetch must land before the LDS is released
#ifdef PP_TIMELINE
    if (p.tokens_in == -777 && blockIdx.x == 0 && lane == 0) {
#pragma unroll
        for (int q = 0; q < 16; ++q) ((unsigned*)p.pos)[wave * 16 + q] = tl_sum[q];
        if (wave == 0) ((unsigned*)p.pos)[128] = (unsigned)(__builtin_amdgcn_s_memtime() - tl_start);      // whole kernel, block 0
    }
#endif
#undef READ_A
#undef READ_B
#undef MMA
#undef BAR
#undef BAR_THEN_WAIT_LDS
}

template <int EPI>
static void launch_pp(const gar_gemm_params& p, int pm, int pn, int num_cus, hipStream_t s) {
    constexpr int LDS = 2 * PSTAGE + (EPI == GAR_EPI_BIAS_GELU ? GELU_LUT_BYTES : 0);
    static gar_once_per_device attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pp_kernel<EPI>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<EPI>), dim3(min(pm * pn, num_cus)), dim3(512), LDS, s, p, pm, pn,
                       gar_gather_args{});                                                                      // persistent
}

static int pp_num_cus() { return gar_num_cus(); }

// Patch-embed + mask-embed convolutions with the patches DMA'd from the image tiles into LDS (see the GATHER notes above).
// x[t, token_offset + patch, :] = [pixel patch | mask patch] Wg^T + pos[token_offset + patch].  Returns GAR_ERR_UNSUPPORTED
// for shapes the gather form is not built for (the caller keeps gar_patch_im2col + gar_gemm).
extern "C" int gar_patch_embed_k(int img, int patch) {
    (void)img;
    return patch > 0 && patch <= 16 ? 2 * GATHER_KT_PER_TENSOR * PBK : 0;
}

extern "C" int gar_patch_embed(int dtype, const void* pixel, const void* maskbin, const void* Wg, const void* pos, void* x,
                               int T_, int img, int patch, int D, int tokens_out, int token_offset, gar_stream_t stream) {
    GAR_CHECK_ARG(pixel && maskbin && Wg && pos && x && T_ > 0 && D > 0, "patch_embed: null pointer / bad shape");
    const int g = patch > 0 ? img / patch : 0;
    const int64_t bytes = (int64_t)T_ * 3 * img * img * 2;
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    const int pm = T_ * g * g / PBM, pn = (D + PBM - 1) / PBM;
    if (dtype != GAR_BF16 || g != 32 || img != g * patch || patch > 16 || (patch & 1) || (patch * img * 2) % 32 != 0 ||
        bytes >= ((int64_t)1 << 32) - 4096 || D % 8 != 0 || D < 256 || pm * pn < 128 || !al16(pixel) || !al16(maskbin) ||
        !al16(Wg) || !al16(pos) || !al16(x) || tokens_out < g * g + token_offset || token_offset < 0) {
        gar_set_error("patch_embed: the gather form is built for bf16, a 32 x 32 grid of even patches <= 16 px, >= 128 output "
                      "tiles (T=%d img=%d patch=%d D=%d)", T_, img, patch, D);
        return GAR_ERR_UNSUPPORTED;
    }
    gar_gemm_params p = {};
    p.A = pixel;
    p.W = Wg;
    p.K = 2 * GATHER_KT_PER_TENSOR * PBK;
    p.ldw = p.K;
    p.lda = p.K;                            // unused by the gather form
    p.C = x;
    p.ldc = D;
    p.M = T_ * g * g;
    p.N = D;
    p.epilogue = GAR_EPI_PATCH_POS;
    p.pos = pos;
    p.tokens_in = g * g;
    p.tokens_out = tokens_out;
    p.token_offset = token_offset;
    gar_gather_args ga;
    ga.mask = maskbin;
    ga.img = img;
    ga.patch = patch;
    ga.bytes = (unsigned)bytes;
    constexpr int LDS = 2 * PSTAGE;
    static gar_once_per_device attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_bf16_pp_kernel<GAR_EPI_PATCH_POS, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    });
    hipLaunchKernelGGL((gemm_bf16_pp_kernel<GAR_EPI_PATCH_POS, true>), dim3(min(pm * pn, pp_num_cus())), dim3(512), LDS,
                       (hipStream_t)stream, p, pm, pn, ga);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// true when the persistent tile GEMM is built for the problem (large bf16 GEMMs; small ones stay on the 128x128 kernel).
// Exported through gar_gemm_tile_takes(): the host asks THIS predicate before it plans a pass around the folded-norm
// epilogues (row_scale / row_stats), which only this kernel has — one copy of the conditions.
bool gar_gemm_pp_takes(const gar_gemm_params& p) {
    const int pm = (p.M + PBM - 1) / PBM, pn = (p.N + PBM - 1) / PBM;
    if (pm * pn < 128 || p.N < 256 || (p.N % 8) != 0) return false;
    // the row-coalesced epilogue moves 16-byte vectors of C / residual / bias / gamma / pos
    auto al16 = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    if ((p.ldc % 8) != 0 || !al16(p.C) || (p.bias && !al16(p.bias)) || (p.gamma && !al16(p.gamma)) ||
        (p.residual && (!al16(p.residual) || (p.ldr % 8) != 0)) || (p.pos && !al16(p.pos)))
        return false;
    // buffer descriptors address each operand with 32-bit byte offsets
    // (num_records and voffset are unsigned 32-bit: operands up to 4 GiB)
    if (((int64_t)(p.M - 1) * p.lda + p.K) * 2 >= ((int64_t)1 << 32) - 4096 ||
        ((int64_t)(p.N - 1) * p.ldw + p.K) * 2 >= ((int64_t)1 << 32) - 4096)
        return false;
    // folded norms: row_scale on the consumer epilogues, row_stats on the producers (include/gar_hip.h)
    const int e_ = p.epilogue;
    if (p.row_scale && !(e_ == GAR_EPI_NONE || e_ == GAR_EPI_BIAS || e_ == GAR_EPI_BIAS_GELU || e_ == GAR_EPI_SWIGLU ||
                         e_ == GAR_EPI_QKV_ROPE_LLM || (e_ == GAR_EPI_QKV_ROPE && !p.qkv_cos && p.qkv_v)))
        return false;
    if (p.row_stats && !(e_ == GAR_EPI_RES || e_ == GAR_EPI_BIAS_SCALE_RES)) return false;
    return e_ >= GAR_EPI_NONE && e_ <= GAR_EPI_QKV_ROPE_LLM;
}

// returns true if the problem was taken
#ifdef GAR_GEMM_LW_VARIANT   /* diagnostic build only (tools/gemm_lw/: the round-6 4-wave 128 x 128-per-wave frame — parity with this kernel, not faster) */
bool gar_gemm_lw_try(const gar_gemm_params& p, hipStream_t s);
#endif

bool gar_gemm_pp_try(const gar_gemm_params& p, hipStream_t s) {
    if (!gar_gemm_pp_takes(p)) return false;
#ifdef GAR_GEMM_LW_VARIANT
    if (gar_gemm_lw_try(p, s)) return true;
#endif
    const int num_cus = pp_num_cus();
    const int pm = (p.M + PBM - 1) / PBM, pn = (p.N + PBM - 1) / PBM;
    switch (p.epilogue) {
        case GAR_EPI_NONE: launch_pp<GAR_EPI_NONE>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_BIAS: launch_pp<GAR_EPI_BIAS>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_BIAS_GELU: launch_pp<GAR_EPI_BIAS_GELU>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_BIAS_SCALE_RES: launch_pp<GAR_EPI_BIAS_SCALE_RES>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_RES: launch_pp<GAR_EPI_RES>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_SWIGLU: launch_pp<GAR_EPI_SWIGLU>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_PATCH_POS: launch_pp<GAR_EPI_PATCH_POS>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_QKV_ROPE: launch_pp<GAR_EPI_QKV_ROPE>(p, pm, pn, num_cus, s); break;
        case GAR_EPI_QKV_ROPE_LLM: launch_pp<GAR_EPI_QKV_ROPE_LLM>(p, pm, pn, num_cus, s); break;
        default: return false;
    }
    return true;
}
