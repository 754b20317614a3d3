// bf16 flash attention, v2 (prefill / ViT): same lane-local S^T -> softmax -> P -> O^T scheme as attention.hip, with
//   * K / Vt tiles DMA'd HBM -> LDS by `buffer_load_dwordx4 ... lds` (no staging VGPRs, no ds_write pass; the XOR
//     swizzle is applied to the per-lane SOURCE offset; rows past the end of the kv slab are zero-filled by the buffer
//     descriptor), double-buffered, the next tile in flight during the whole compute of the current one;
//   * raw v_exp_f32, first MFMA of every accumulator chain fed with an inline-zero C operand;
//   * deferred rescale: O / l are rescaled only when some row's running max grew by more than 2^RESCALE_THR.
#include <stdlib.h>

#include "common.h"

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define RESCALE_THR 6.0f   // log2 domain

#ifdef ATTN_TIMELINE   /* diagnostic build (tools/attn_timeline.py): shader-clock stamps between the segments of the kv loop;
                          the (b = 0, head = 0, middle q-block) workgroup writes its per-wave sums over rows 0..3 of O */
#define TLA(i) { __builtin_amdgcn_sched_barrier(0); tl_t[i] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#else
#define TLA(i)
#endif

typedef float f32v2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk(float lo, float hi) { return pack_bf2(lo, hi); }

__device__ __forceinline__ f32x16 zero16_c() {
    return f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
}

// 2^x for x <= 0 on the FMA pipe, two elements per instruction where a packed form exists: t = x + 1.5 * 2^23 rounds x to the
// nearest integer n in t's low mantissa bits; f = x - n in [-0.5, 0.5]; 2^f by a degree-3 polynomial (|rel err| < 1.1e-4, bf16
// rounds P at 3.9e-3); the result's exponent += n by an integer add of (bits(t) << 23).
__device__ __forceinline__ f32v2_t exp2_poly2(f32v2_t x) {
    x[0] = __builtin_amdgcn_fmed3f(x[0], -125.0f, 0.0f);
    x[1] = __builtin_amdgcn_fmed3f(x[1], -125.0f, 0.0f);
    const f32v2_t magic = {12582912.0f, 12582912.0f};
    const f32v2_t t = x + magic;
    const f32v2_t n = t - magic;
    const f32v2_t f = x - n;
    const f32v2_t c3 = {0.05590050f, 0.05590050f}, c2 = {0.24015480f, 0.24015480f}, c1 = {0.69313330f, 0.69313330f},
                  one = {1.0f, 1.0f};
    f32v2_t p = __builtin_elementwise_fma(f, c3, c2);
    p = __builtin_elementwise_fma(p, f, c1);
    p = __builtin_elementwise_fma(p, f, one);
    f32v2_t r;
    r[0] = __uint_as_float(__float_as_uint(p[0]) + (__float_as_uint(t[0]) << 23));
    r[1] = __uint_as_float(__float_as_uint(p[1]) + (__float_as_uint(t[1]) << 23));
    return r;
}

template <int RS> __device__ __forceinline__ int key_of(int row) { return RS == 128 ? ((row >> 1) & 7) : (row & 15); }

// VROW: V arrives row-major [kv][HD] — the layout the fused qkv GEMM epilogue writes, no transpose pass — is
// DMA'd like a K tile, and the PV A operand (rows = d, k = kv: column-major in that image) comes out of
// ds_read_b64_tr_b16, gfx950's transposing LDS read: each 16-lane group reads one [4 kv][16 d] block, lane i supplying
// the address of the block's chunk (kv row i >> 2, d columns 4 (i & 3) ..) and receiving column i; two reads = the 8
// consecutive kv of one d that the MFMA fragment wants. The 16-byte chunks of a V row are XOR-ed with 4 ((kv >> 1) & 1):
// the four rows of a block (128 B apart) then cover all 64 banks exactly once for a 32-lane half.
typedef short tr4_t __attribute__((ext_vector_type(4)));
#ifndef ATTN_POLY_EXP      /* n > 0 (even): the first n of every 16 scores of a block are exponentiated on the FMA pipe — Cody-Waite range
                              reduction (magic-number rounding) + degree-3 polynomial in packed fp32 + exponent add — instead of
                              v_exp_f32, so that the transcendental unit and the FMA pipe share the softmax (VERDICT r3 #1b) */
#define ATTN_POLY_EXP 0
#endif
#ifndef ATTN_KPRE          /* 1: the tile's K fragments are all read before the first QK^T MFMA */
#define ATTN_KPRE 0
#endif
#ifndef ATTN_VPRE          /* 1: the tile's V fragments are read before the softmax arithmetic (VROW only), consumed after it */
#define ATTN_VPRE 0
#endif
#ifndef ATTN_PK_SUM        /* n > 0: row sums of P in n packed-fp32 partial chains (v_pk_add_f32) instead of 32 scalar adds per tile */
#define ATTN_PK_SUM 0
#endif
#ifndef ATTN_MFMA_ROWSUM   /* 1: row sums of P on the matrix pipe (ones x P), instead of 32 VALU adds per tile and lane */
#define ATTN_MFMA_ROWSUM 0
#endif
template <int HD, bool CAUSAL, bool VROW = false>
__global__ __launch_bounds__(256, 2) void attn_bf16_v2_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                              int Hq, int Hkv, int q_len, int q_pad, int kv_len_arg,
                                                              int kv_stride, const int32_t* __restrict__ kv_len_dev,
                                                              const int32_t* __restrict__ kv_start, int kv_prefix, int o_rows) {
    // head_dim 96 (PE-G/14): K rows are 192 B in HBM; the LDS image keeps the 256-B row pitch of head_dim 128 (a
    // 4-row x 256-B DMA piece per instruction; the 64 bytes past a row's 12 real chunks are filled with a repeat of
    // chunk 11 and never read), so the fragment reads use the conflict-free head_dim-128 swizzle
    constexpr int KRS_G = HD * 2;               // K row bytes in HBM
    constexpr int KRS = HD == 64 ? 128 : 256;   // K row bytes in LDS
    constexpr int KT = 64 * KRS;                // K tile bytes in LDS   [64 kv][KRS / 2]
    constexpr int VT = VROW ? KT : HD * 128;    // Vt tile bytes [HD][64 kv]; VROW: a V tile has the shape and LDS pitch of a K tile
    constexpr int NKD = HD / 16;                // QK^T k-steps
    constexpr int NDB = HD / 32;                // O^T row blocks
    constexpr int KI = KT / 4096;               // K DMA instructions per wave per tile (1 KiB each)
    constexpr int VI = VT / 4096;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K | Vt]
    // kv_prefix = 1 (non-causal, row-major V: the ViT with its cls token): key / value row 0 is folded into the INITIAL
    // softmax state instead of riding in a kv tile — m0 = q . k0, l0 = 1, O0 = v0 — and the tiles cover rows 1 .. kv_len - 1:
    // 1025 keys are 16 full tiles instead of 17 with one live column in the last (and no masked tile at all).
    const int PFX = (!CAUSAL && VROW) ? kv_prefix : 0;
    const int kv_len = (kv_len_dev ? kv_len_dev[0] : kv_len_arg) - PFX;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // 1-D grid, XCD-aware: workgroups go round-robin over the 8 XCDs (each with its own L2), so workgroup L runs on XCD
    // L % 8. Give every XCD one contiguous chunk of the (batch, head, q-block) list: the q-blocks of a (batch, head) —
    // and the Hq/Hkv query heads that share a kv head — then run on the same XCD at the same time and re-read their
    // K / Vt tiles from that L2 instead of each fetching them over the fabric (9x / 148x re-fetch otherwise: measured
    // ~5.8 TB/s of fabric traffic, the kernel was bound by it).
    const int nqb = (q_len + 127) >> 7;
    int wk;
    {
        const int total = gridDim.x, L = blockIdx.x;
        const int per = total >> 3, rem = total & 7, xcd = L & 7, slot = L >> 3;
        wk = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    const int qb = nqb - 1 - (wk % nqb);        // heavy (late) causal blocks first
    const int head = (wk / nqb) % Hq, b = wk / (nqb * Hq);
    const int kvh = head / (Hq / Hkv);
    const int q0 = qb * 128 + wave * 32;        // this wave's first query
    const int coff = kv_len - q_len;            // causal: kv <= q + coff
    const bf16_t* Qp = Q + (((int64_t)b * Hq + head) * q_pad) * HD;
    const bf16_t* Kp = K + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    const bf16_t* Vp = VROW ? Vt + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD
                            : Vt + (((int64_t)b * Hkv + kvh) * HD) * (int64_t)kv_stride;
    const unsigned slab = (unsigned)kv_stride * HD * 2u;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)slab, 0x00020000);

    // per-lane source offsets of DMA piece 0 (piece i adds a scalar): LDS image is lane-linear, 1 KiB per piece
    int voffK, voffV;
    if (HD == 64) {          // piece = 8 rows x 128 B
        const int row = wave * 8 + (lane >> 3);
        voffK = row * 128 + (((lane & 7) ^ key_of<128>(row)) << 4);
    } else {                 // piece = 4 rows x 256 B of LDS
        const int row = wave * 4 + (lane >> 4);
        const int cl = (lane & 15) ^ key_of<256>(row);                    // logical 16-byte chunk this lane fetches
        voffK = row * KRS_G + (min(cl, KRS_G / 16 - 1) << 4);
    }
    if (VROW && HD == 64) { // V piece = 8 kv rows x 128 B (64 d); chunk key 4 ((row >> 1) & 1)
        const int row = wave * 8 + (lane >> 3);
        voffV = row * 128 + (((lane & 7) ^ (((row >> 1) & 1) << 2)) << 4);
    } else if (VROW) {      // V piece = 4 kv rows x 256 B of LDS (like K); chunk key 4 (row & 3): the four rows of a tr block
        const int row = wave * 4 + (lane >> 4);                          // are 256 B apart and take the four 64-byte bank groups
        const int cl = (lane & 15) ^ ((row & 3) << 2);
        voffV = row * KRS_G + (min(cl, KRS_G / 16 - 1) << 4);
    } else {
        const int row = wave * 8 + (lane >> 3);      // Vt piece = 8 d-rows x 128 B (64 kv)
        voffV = (int)((unsigned)row * (unsigned)kv_stride * 2u) + (((lane & 7) ^ key_of<128>(row)) << 4);
    }
    auto stage = [&](int t, int buf) {
        char* ks = smem + buf * (KT + VT);
        char* vs = ks + KT;
        const unsigned kbase = ((unsigned)t * 64u + (unsigned)PFX) * KRS_G;               // 64 kv rows per tile
        const unsigned vbase = VROW ? ((unsigned)t * 64u + (unsigned)PFX) * KRS_G : (unsigned)t * 128u;     // 64 kv rows / 64 kv columns
#pragma unroll
        for (int i = 0; i < KI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, LDS_AS(ks + (i * 4 + wave) * 1024), 16,
                                                     voffK + (int)(kbase + (unsigned)i * (4096u / KRS) * KRS_G), 0, 0, 0);
#pragma unroll
        for (int i = 0; i < VI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_AS(vs + (i * 4 + wave) * 1024), 16,
                                                     voffV + (int)(vbase + (VROW ? (unsigned)i * (4096u / KRS) * KRS_G
                                                                                         : (unsigned)i * 32u * (unsigned)kv_stride * 2u)),
                                                     0, 0, 0);
    };

    // Q fragments (B operand): Q[q0 + l31][16 kd + 8h .. +8]
    bf16x8 qf[NKD];
    {
        const int qrow = min(q0 + l31, q_pad - 1);
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
            qf[kd] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)qrow * HD + kd * 16 + h * 8);
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    if (PFX) {
        // s0 = q . k0 (q carries scale * log2e): this lane holds dims 16 kd + 8 h .. + 8 of its query row
        float part = 0.f;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) {
            const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Kp + kd * 16 + h * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                part = __builtin_fmaf(bf2f((bf16_t)qf[kd][e]), bf2f((bf16_t)kf[e]), part);
        }
        m_run = part + __shfl_xor(part, 32, 64);
        l_run = h == 0 ? 1.0f : 0.f;                // the two halves' l are added at the end
        // O0 = v0: register r of row block d is d index 32 d + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v4[4];
                ld4(Vp + d * 32 + g * 8 + h * 4, v4);
#pragma unroll
                for (int r = 0; r < 4; ++r) o[d][g * 4 + r] = v4[r];
            }
    }
    // The QK^T accumulator chains start from -m (the running max the scores are measured against) instead of 0: the MFMA
    // delivers s - m and the softmax needs no subtraction — 32 VALU instructions fewer per kv tile in a loop whose SIMD
    // time is matrix pipe time plus most of the VALU issue time (tools/coissue_probe.hip, profiles/r2_attention_timeline.txt).
    // 16 VGPRs, rewritten only when the max is re-based. With the exact max on every tile this paid for non-causal
    // head_dim 64 only (the causal instantiation went from 144 to 201 VGPRs: -7 %); with the lazy max below the -m vector
    // is all but constant and every instantiation gains: causal head_dim 64 +5.4 %, head_dim 128 +4.5 %, 96 +4 %
    // (same-box A/Bs of -DATTN_NEGM_SET builds, profiles/r2_attention_lazy_max.txt).
#ifndef ATTN_NEGM_SET       /* bit 0: head_dim 64, bit 1: 128, bit 2: 96 */
#define ATTN_NEGM_SET 7
#endif
    constexpr bool NEGM = (HD == 64 && (ATTN_NEGM_SET & 1)) || (HD == 128 && (ATTN_NEGM_SET & 2)) || (HD == 96 && (ATTN_NEGM_SET & 4));
    f32x16 negm = zero16_c();
    if (PFX && NEGM) {
#pragma unroll
        for (int r = 0; r < 16; ++r) negm[r] = -m_run;
    }

    // Left-padded batch (CAUSAL only; kv_start[b] = first real position of sequence b, NULL = 0): a key is visible iff
    // kv_lo <= kv <= max(q + coff, kv_lo). Rows in front of kv_lo (padding queries, never read) see exactly key kv_lo, so every
    // row has a visible key in the first tile it visits — the tile holding kv_lo — and the running max is finite from there.
    const int kv_lo = kv_start ? max(min(kv_start[b], kv_len - 1), 0) : 0;
    const int t_lo = kv_lo >> 6;
    int kv_end = kv_len;
    if (CAUSAL) kv_end = min(kv_len, max(qb * 128 + 127 + coff, kv_lo) + 1);
    const int ntiles = (kv_end + 63) / 64;
    if (ntiles > t_lo) stage(t_lo, t_lo & 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2<->3 swapped
    const bool wave_active = q0 < q_len;
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // LDS byte offsets of this lane's fragment rows
    int koff[2], kkey[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int row = blk * 32 + prow;
        koff[blk] = row * KRS;
        kkey[blk] = key_of<KRS>(row);
    }
    // VROW: byte offset inside a V tile of this lane's tr-read chunk for d-block 0 (d-block 1 = chunk index + 4 = ^ 64 bytes)
    int vtr = 0;
    if (VROW) {
        const int i = lane & 15, r = 8 * h + (i >> 2);                    // kv row inside the 16-kv step
        const int col = 16 * ((lane >> 4) & 1) + 4 * (i & 3);             // d column inside the 32-d block
        const int key4 = HD == 64 ? ((r >> 1) & 1) << 2 : (r & 3) << 2;
        vtr = r * KRS + ((((col >> 3) ^ key4) << 4) | ((col & 7) << 1));
    }
#ifdef ATTN_TIMELINE
    unsigned tl_t[8], tl_sum[8];
    for (int i = 0; i < 8; ++i) tl_sum[i] = 0u;
    unsigned tl_n = 0u;
    const unsigned tl_begin = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    for (int t = t_lo; t < ntiles; ++t) {
        const int buf = t & 1;
        TLA(0)
        if (t + 1 < ntiles) stage(t + 1, buf ^ 1);
        TLA(1)
        const int kv0 = t * 64;
        const bool skip = !wave_active || (CAUSAL && kv0 > max(q0 + 31 + coff, kv_lo));     // wave-uniform
        if (!skip) {
            const char* ks = smem + buf * (KT + VT);
            const char* vs = ks + KT;
            f32x16 s[2];
            auto qk = [&]() {
#if ATTN_KPRE
                // all K fragments of the tile in flight before the first MFMA (the compiler otherwise re-uses one register quad
                // and waits for every ds_read in front of its MFMA: eight exposed LDS latencies per tile and wave)
                bf16x8 kfa[2][NKD];
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int kd = 0; kd < NKD; ++kd)
                        kfa[blk][kd] = *reinterpret_cast<const bf16x8*>(ks + koff[blk] + (((kd * 2 + h) ^ kkey[blk]) << 4));
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kd = 0; kd < NKD; ++kd)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
                        s[blk] = MFMA_32x32x16(kfa[blk][kd], qf[kd], kd == 0 ? (NEGM ? negm : zero16) : s[blk]);
#else
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                    for (int kd = 0; kd < NKD; ++kd) {
                        const bf16x8 kf =
                            *reinterpret_cast<const bf16x8*>(ks + koff[blk] + (((kd * 2 + h) ^ kkey[blk]) << 4));
                        s[blk] = MFMA_32x32x16(kf, qf[kd], kd == 0 ? (NEGM ? negm : zero16) : s[blk]);
                    }
                }
#endif
            };
            qk();
#if ATTN_VPRE
            bf16x8 vfa[NDB][2][2];
            if (VROW) {
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                        for (int tt = 0; tt < 2; ++tt) {
                            const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) tr4_t*)(vs + (blk * 32 + tt * 16) * KRS + (vtr ^ (d << 6))));
                            const tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) tr4_t*)(vs + (blk * 32 + tt * 16 + 4) * KRS + (vtr ^ (d << 6))));
                            vfa[d][blk][tt] = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        }
                __builtin_amdgcn_sched_barrier(0);
            }
#endif
            TLA(2)
            // register r of block blk <-> kv = kv0 + 32 blk + 16 (r>>3) + 8 h + (r&7)
            const bool need_mask = (kv0 + 64 > kv_len) || kv0 < kv_lo || (CAUSAL && kv0 + 63 > q0 + coff);   // wave-uniform
            float ps = 0.f;
            bf16x8 pf[2][2];
            // p = exp2(s - m_sub) (NEGM: s already carries -m), row sum, bf16 pack
            auto exp_pack = [&](float m_sub) {
                ps = 0.f;
#if ATTN_PK_SUM
                f32v2_t ps2[ATTN_PK_SUM] = {};          // row sums as packed fp32 adds (v_pk_add_f32: two adds per instruction)
#endif
#pragma unroll
                for (int blk = 0; blk < 2; ++blk) {
                    float p[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
#if ATTN_POLY_EXP
                        if (r < ATTN_POLY_EXP) {
                            if ((r & 1) == 0) {
                                const f32v2_t e2 = exp2_poly2(f32v2_t{NEGM ? s[blk][r] : s[blk][r] - m_sub,
                                                                      NEGM ? s[blk][r + 1] : s[blk][r + 1] - m_sub});
                                p[r] = e2[0];
                                p[r + 1] = e2[1];
                            }
                        } else
#endif
                        p[r] = __builtin_amdgcn_exp2f(NEGM ? s[blk][r] : s[blk][r] - m_sub);
#if !ATTN_PK_SUM
                        if (!ATTN_MFMA_ROWSUM) ps += p[r];
#endif
                    }
#if ATTN_PK_SUM
#pragma unroll
                    for (int r = 0; r < 16; r += 2) ps2[(r >> 1) % ATTN_PK_SUM] += f32v2_t{p[r], p[r + 1]};
#endif
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        u32x4 w;
                        w[0] = cvt_pk(p[tt * 8 + 0], p[tt * 8 + 1]);
                        w[1] = cvt_pk(p[tt * 8 + 2], p[tt * 8 + 3]);
                        w[2] = cvt_pk(p[tt * 8 + 4], p[tt * 8 + 5]);
                        w[3] = cvt_pk(p[tt * 8 + 6], p[tt * 8 + 7]);
                        pf[blk][tt] = __builtin_bit_cast(bf16x8, w);
                    }
                }
#if ATTN_PK_SUM
                {
                    f32v2_t t2 = ps2[0];
#pragma unroll
                    for (int i = 1; i < ATTN_PK_SUM; ++i) t2 += ps2[i];
                    ps = t2[0] + t2[1];
                }
#endif
                if (ATTN_MFMA_ROWSUM) {
                    // ones[32 x 16] x P[16 kv x 32 q], four k-steps: every register of the result is the COMPLETE row sum of
                    // this lane's query column (both kv halves) of the bf16-rounded P — what the PV product divides by
                    const bf16x8 ones = {H16_ONE, H16_ONE, H16_ONE, H16_ONE, H16_ONE, H16_ONE, H16_ONE, H16_ONE};
                    f32x16 ls = MFMA_32x32x16(ones, pf[0][0], zero16);
                    ls = MFMA_32x32x16(ones, pf[0][1], ls);
                    ls = MFMA_32x32x16(ones, pf[1][0], ls);
                    ls = MFMA_32x32x16(ones, pf[1][1], ls);
                    ps = ls[0];
                }
            };
            // Lazy running max. The max only guards the exponent range: P and the O / l accumulators are rounded
            // relative to their own magnitude, so measuring a tile's scores against a max that is up to 2^LAZY_LOG2 too
            // small costs no accuracy. Tile 0 (and every masked tile) takes the exact path below and leaves a finite max
            // under which the row's l is >= 1; every later tile exponentiates against that max at once — no 32-way max,
            // no half-row exchange, no compare — and only looks at the row sums it needs anyway: a lane whose partial sum
            // reaches 2^LAZY_LOG2 (or is inf / NaN) sends the wave through the exact path, which re-bases the max and
            // redoes the tile from the scores still in registers.
#ifndef ATTN_LAZY_LOG2
#define ATTN_LAZY_LOG2 H16_MAX_LOG2       /* 0: always the exact path (A/B); 16, or 15 where P is stored as fp16 */
#endif
            bool exact = ATTN_LAZY_LOG2 == 0 || t == t_lo || need_mask;                       // wave-uniform
            if (!exact) {
                exp_pack(NEGM ? 0.f : m_run);
                exact = !__all(ps < (float)(1u << ATTN_LAZY_LOG2));
            }
            if (exact) {
            if (need_mask) {
                const int qi = q0 + l31;
                const int lim = CAUSAL ? min(kv_len - 1, max(qi + coff, kv_lo)) : kv_len - 1;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + blk * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                        s[blk][r] = (kv <= lim && kv >= kv_lo) ? s[blk][r] : -INFINITY;
                    }
            }
            float mx = -INFINITY;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));     // relative to m_base = -negm (0 while m_run is still -inf)
            TLA(3)
            float m_use = 0.f;                  // what is still to be subtracted from the scores
            if (!NEGM) {
                if (!__all(mx - m_run <= RESCALE_THR)) {           // NaN (-inf - -inf) also lands here
                    const float m_new = fmaxf(m_run, mx);
                    const float m_nu = m_new == -INFINITY ? 0.f : m_new;
                    const float alpha = __builtin_amdgcn_exp2f(m_run - m_nu);
                    m_run = m_new;
                    l_run *= alpha;
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
#pragma unroll
                        for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
                }
                m_use = m_run == -INFINITY ? 0.f : m_run;
            } else
            // (m_base - m_run) is 0 once the row has a finite max and +inf before: NaN / +inf also land in the branch
            if (!__all(mx + (-negm[0] - m_run) <= RESCALE_THR)) {
                const float m_base = -negm[0];
                const float m_new = fmaxf(m_run, mx + m_base);
                const float m_nu = m_new == -INFINITY ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_nu);
                const float shift = m_base - m_nu;              // re-base this tile's scores on the new max
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[blk][r] += shift;
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[r] = -m_nu;
            }
            exp_pack(m_use);
            }
            l_run += ps;
            TLA(4)
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                const int row = d * 32 + l31;
                const int key = key_of<128>(row);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        bf16x8 vf;
#if ATTN_VPRE
                        if (VROW) {
                            vf = vfa[d][blk][tt];
                        } else
#endif
                        if (VROW) {
                            // this lane's chunk of its group's [4 kv][16 d] block: kv row (blk*32 + tt*16 + 8h) + (i >> 2) (+4),
                            // d columns d*32 + 16 ((lane >> 4) & 1) + 4 (i & 3);  i = lane & 15.  vtr = byte offset of (kv row i >> 2 of
                            // the wave-tile's row 8h, that d chunk) with the row key of rows 0-1; rows 2-3 flip chunk bit 2
                            const tr4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) tr4_t*)(vs + (blk * 32 + tt * 16) * KRS + (vtr ^ (d << 6))));
                            const tr4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                                (__attribute__((address_space(3))) tr4_t*)(vs + (blk * 32 + tt * 16 + 4) * KRS + (vtr ^ (d << 6))));
                            vf = bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        } else {
                            const int c = (blk * 2 + tt) * 2 + h;
                            vf = *reinterpret_cast<const bf16x8*>(vs + row * 128 + ((c ^ key) << 4));
                        }
                        o[d] = MFMA_32x32x16(vf, pf[blk][tt], o[d]);
                    }
            }
            TLA(5)
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        TLA(6)
        __syncthreads();
        TLA(7)
#ifdef ATTN_TIMELINE
        if (!skip && !(CAUSAL && kv0 + 63 > q0 + coff) && kv0 + 64 <= kv_len && kv0 >= kv_lo) {     // full, unmasked tiles only
            for (int i = 0; i < 7; ++i) tl_sum[i] += tl_t[i + 1] - tl_t[i];
            ++tl_n;
        }
#endif
    }
#ifdef ATTN_TIMELINE
    {
        const unsigned tl_total = (unsigned)__builtin_amdgcn_s_memtime() - tl_begin;
        const bool writer = b == 0 && head == 0 && qb == nqb / 2;
        if (writer && lane < 16) {
            unsigned v = 0u;                   // dynamic register indexing would go to scratch: select with a chain
            for (int i = 0; i < 7; ++i) v = lane == i ? tl_sum[i] : v;
            v = lane == 7 ? tl_n : lane == 8 ? tl_total : lane == 9 ? (unsigned)ntiles : v;
            reinterpret_cast<unsigned*>(O + (int64_t)wave * ((int64_t)Hq * HD))[lane] = v;
        }
        if (b == 0 && head == 0 && qb == 0) return;      // the rows the writer uses
    }
#endif
    // epilogue: O[b*q_len + q][head*HD + d], d = 32 db + (r&3) + 8 (r>>2) + 4 h
    const float l_tot = ATTN_MFMA_ROWSUM ? l_run : l_run + __shfl_xor(l_run, 32, 64);
    const int qi = q0 + l31;
    if (qi < q_len) {
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        bf16_t* op = O + ((int64_t)b * o_rows + qi) * ((int64_t)Hq * HD) + head * HD;      // o_rows: rows of O per batch item (>= q_len)
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {o[d][g * 4 + 0] * inv, o[d][g * 4 + 1] * inv, o[d][g * 4 + 2] * inv, o[d][g * 4 + 3] * inv};
                st4(op + d * 32 + g * 8 + h * 4, v);
            }
    }
}

// returns false when this kernel does not apply (caller falls back to the register-staged kernel of attention.hip).
// vrow: V is row-major [B, Hkv, kv_stride, hd] instead of transposed (head_dim 64 only).
bool gar_attn_bf16_v2_try(const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv, int hd, int q_len,
                          int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev, int vrow,
                          const int32_t* kv_start, int kv_prefix, hipStream_t s, int o_rows) {
    if ((int64_t)kv_stride * hd * 2 >= (int64_t)1 << 31) return false;
    if (o_rows <= 0) o_rows = q_len;          // the first q_len query rows of sequences that hold o_rows rows in O (v4 takes the rest)
    dim3 grid(((q_len + 127) / 128) * Hq * B), block(256);
    const int kt = 64 * (hd == 64 ? 128 : 256);
    const int lds = 2 * (kt + (vrow ? kt : hd * 128));
#define LAUNCH_V2(HD_, C_, V_)                                                                                         \
    hipLaunchKernelGGL((attn_bf16_v2_kernel<HD_, C_, V_>), grid, block, lds, s, (const bf16_t*)Q, (const bf16_t*)K,    \
                       (const bf16_t*)Vt, (bf16_t*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start, kv_prefix, o_rows)
    if (hd == 64) {
        if (vrow) { if (causal) LAUNCH_V2(64, true, true); else LAUNCH_V2(64, false, true); }
        else { if (causal) LAUNCH_V2(64, true, false); else LAUNCH_V2(64, false, false); }
    }
    else if (hd == 128) {
        if (vrow) { if (causal) LAUNCH_V2(128, true, true); else LAUNCH_V2(128, false, true); }
        else { if (causal) LAUNCH_V2(128, true, false); else LAUNCH_V2(128, false, false); }
    } else if (hd == 96) {
        if (vrow) { if (causal) LAUNCH_V2(96, true, true); else LAUNCH_V2(96, false, true); }
        else { if (causal) LAUNCH_V2(96, true, false); else LAUNCH_V2(96, false, false); }
    }
    else return false;
#undef LAUNCH_V2
    return true;
}
