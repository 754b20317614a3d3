// Decode attention (q_len = 1): split-KV, GQA-aware, HBM-bound.
//
//   Block = 4 waves = one (split, kv head, batch). The G = Hq/Hkv query heads sharing a kv head are the columns of
//   the MFMA B operand (columns >= G are zero), so K and V of that kv head stream from HBM exactly ONCE per step.
//   The block's kv range is cut in 64-kv tiles dealt round-robin to its waves; each wave brings its K / V tiles — both
//   row-major [kv][HD], the cache's layout — into a private LDS region with `buffer_load ... lds` DMA, forms the QK^T
//   fragments with ds_read_b128 and the PV fragments (k = kv: a column of the V tile) with gfx950's transposing
//   ds_read_b64_tr_b16, keeps an online softmax, and the four waves are merged in LDS into one un-normalised partial
//   (m, l, O[G][HD]) per block. decode_combine_kernel merges the splits.
//   kv length comes from device memory so the captured hipGraph replays unchanged for every token.
//   Algorithmic bytes per launch: B * Hkv * kv_len * HD * 2 (K and V) * sizeof(bf16).
#include <stdlib.h>

#include "common.h"

typedef float f32v2_d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk_d(float lo, float hi) { return pack_bf2(lo, hi); }

typedef short tr4_d __attribute__((ext_vector_type(4)));
// PV A-operand fragment (rows = d, k = 8 consecutive kv) out of a row-major V tile in LDS: two transposing reads, each a
// [4 kv][16 d] block per 16-lane group (see attention_bf16.hip, VROW); `p` = this lane's chunk address for kv rows 0..3
__device__ __forceinline__ bf16x8 v_frag_tr(const char* p, int row4_bytes) {
    const tr4_d lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4_d*)(p));
    const tr4_d hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) tr4_d*)(p + row4_bytes));
    return bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}

// ---------------------------------------------------------------------------------------------------------------
// head_dim 64: every wave brings its 64-kv K / V tiles in with `buffer_load_dwordx4 ... lds` — 1 KiB per instruction,
// eight full 128-byte lines (per-lane fragment loads from global memory touch 32 different lines, 32 B each, per
// instruction: measured with the compute removed they top out at 3.3 TB/s). A wave owns a private 16 KiB LDS region
// (K tile | V tile, XOR-swizzled through the DMA source offsets like attention_bf16.hip): wait for tile t, pull its
// fragments into registers, immediately re-arm the region with the DMA of the wave's next tile, then compute on the
// registers — no block-level barrier in the loop.
// 64 KiB + merge buffer per block -> two blocks (8 waves) per CU, 128 KiB of loads in flight per CU.
// ---- gar_llm_qkv_post (S = 1) fused into the attention launch (gar_attention_decode_qkv) ------------------------------
// The step's raw qkv GEMM output [B, (Hq + 2 Hkv) HD] replaces Q: every block rotates and scales its G query heads while it
// loads them (a lane's MFMA fragments hold dims 16 kd + 8 h + e for all kd: both halves of every rotation pair), and the
// wave that owns the LAST kv tile rotates the new key, appends key and value to the caches (global rows `pos`) and patches
// them into its LDS tiles after the DMA of that tile has landed — the DMA may or may not have seen the global rows.
struct decode_qkv_args {
    const bf16_t* raw;          // NULL: plain gar_attention_decode (Q, caches already appended)
    const float* cs;
    const float* sn;            // [max_pos, HD / 2] f32
    float q_scale;
    int strip;                  // q / k head columns of `raw` in the fused RoPE epilogue's strip order (head_dim 128: dims
                                // [0..31, 64..95, 32..63, 96..127]; include/gar_hip.h GAR_EPI_QKV_ROPE_LLM) — the folded qkv weight's
};
template <int HD>
__device__ __forceinline__ int strip_col_d(int d0, bool strip) {
    if (HD != 128 || !strip) return d0;
    const int blk = d0 >> 5;
    return blk == 1 ? d0 + 32 : (blk == 2 ? d0 - 32 : d0);
}

template <int HD>
__device__ __forceinline__ void load_q_rotated(bf16x8 (&qf)[HD / 16], const decode_qkv_args& fa, int b, int Hq, int Hkv, int head,
                                               bool valid, int h, int rp) {
    constexpr int HALF = HD / 2, NKD = HD / 16;
    const bf16_t* rq = fa.raw + ((int64_t)b * (Hq + 2 * Hkv) + head) * HD;
#pragma unroll
    for (int kd = 0; kd < NKD / 2; ++kd) {
        const int off = kd * 16 + h * 8;
        float x1[8], x2[8], c[8], sv[8];
        ld8(rq + strip_col_d<HD>(off, fa.strip), x1);
        ld8(rq + strip_col_d<HD>(HALF + off, fa.strip), x2);
        ld8(fa.cs + (int64_t)rp * HALF + off, c);
        ld8(fa.sn + (int64_t)rp * HALF + off, sv);
#pragma unroll
        for (int e = 0; e < 8; ++e) rope_half_pair(x1[e], x2[e], c[e], sv[e], fa.q_scale, x1[e], x2[e]);
        const u32x4 w1 = {pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3]), pack_bf2(x1[4], x1[5]), pack_bf2(x1[6], x1[7])};
        const u32x4 w2 = {pack_bf2(x2[0], x2[1]), pack_bf2(x2[2], x2[3]), pack_bf2(x2[4], x2[5]), pack_bf2(x2[6], x2[7])};
        const u32x4 z = {0u, 0u, 0u, 0u};
        qf[kd] = __builtin_bit_cast(bf16x8, valid ? w1 : z);
        qf[kd + NKD / 2] = __builtin_bit_cast(bf16x8, valid ? w2 : z);
    }
}

// new key (rotated) and value row of (b, kvh) at cache row `pos`: global append + patch of the wave's LDS tiles (row pos & 63).
// KEYFN / VKEYFN: chunk swizzle of a K / V tile row (the kernels' DMA layouts). Lanes 0 .. HD/16-1 do the key (8 dims of each
// half), lanes 0 .. HD/8-1 the value.
template <int HD, bool DO_K, bool DO_V, typename KF, typename VF>
__device__ __forceinline__ void append_new_kv(const decode_qkv_args& fa, bf16_t* Kp, bf16_t* Vp, char* ks, char* vs, int b, int Hq,
                                              int Hkv, int kvh, int lane, int pos, int rp, KF kkey, VF vkey) {
    constexpr int HALF = HD / 2, RB = HD * 2;
    const int r = pos & 63;
    const bf16_t* row = fa.raw + (int64_t)b * (Hq + 2 * Hkv) * HD;
    if (DO_K && lane < HALF / 8) {
        const int i8 = lane * 8;
        const bf16_t* rk = row + (Hq + kvh) * HD;
        float x1[8], x2[8], c[8], sv[8];
        ld8(rk + strip_col_d<HD>(i8, fa.strip), x1);
        ld8(rk + strip_col_d<HD>(HALF + i8, fa.strip), x2);
        ld8(fa.cs + (int64_t)rp * HALF + i8, c);
        ld8(fa.sn + (int64_t)rp * HALF + i8, sv);
#pragma unroll
        for (int e = 0; e < 8; ++e) rope_half_pair(x1[e], x2[e], c[e], sv[e], 1.0f, x1[e], x2[e]);
        const uint4 w1 = make_uint4(pack_bf2(x1[0], x1[1]), pack_bf2(x1[2], x1[3]), pack_bf2(x1[4], x1[5]), pack_bf2(x1[6], x1[7]));
        const uint4 w2 = make_uint4(pack_bf2(x2[0], x2[1]), pack_bf2(x2[2], x2[3]), pack_bf2(x2[4], x2[5]), pack_bf2(x2[6], x2[7]));
        *reinterpret_cast<uint4*>(Kp + (int64_t)pos * HD + i8) = w1;
        *reinterpret_cast<uint4*>(Kp + (int64_t)pos * HD + HALF + i8) = w2;
        *reinterpret_cast<uint4*>(ks + r * RB + ((lane ^ kkey(r)) << 4)) = w1;
        *reinterpret_cast<uint4*>(ks + r * RB + (((lane + HALF / 8) ^ kkey(r)) << 4)) = w2;
    }
    if (DO_V && lane < HD / 8) {
        const uint4 w = *reinterpret_cast<const uint4*>(row + (Hq + Hkv + kvh) * HD + lane * 8);
        *reinterpret_cast<uint4*>(Vp + (int64_t)pos * HD + lane * 8) = w;
        *reinterpret_cast<uint4*>(vs + r * RB + ((lane ^ vkey(r)) << 4)) = w;
    }
    asm volatile("" ::: "memory");          // the fragment reads that follow (ds_read / transposing reads) stay behind the patch
}

#ifndef DA_KV_AUX               /* cache policy of the K / V DMA: 2 = nt — every byte of the cache is read once per step:
                                   B = 64, kv 4750: 105.8 -> 98.2 us (5.88 -> 6.34 TB/s; profiles/r3_decode_nt.txt) */
#define DA_KV_AUX 2
#endif
#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
__global__ __launch_bounds__(256, 2) void decode_attn_lds_kernel(const bf16_t* __restrict__ Q, bf16_t* K, bf16_t* V,      // K / V: appended to in fused mode
                                                                 float* __restrict__ part,
                                                                 int Hq, int Hkv, int kv_stride,
                                                                 const int32_t* __restrict__ kv_len_dev, const int32_t* __restrict__ kv_start,
                                                                 bf16_t* __restrict__ O_direct, const decode_qkv_args fa, int64_t q_stride) {
    constexpr int HD = 64, NKD = HD / 16, NDB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [4 waves][K 8 KiB | V 8 KiB] then red
    float (*red)[8][HD + 2] = reinterpret_cast<float (*)[8][HD + 2]>(smem + 4 * 16384);
    const int kv_len = kv_len_dev[0] + (fa.raw ? 1 : 0);            // fused: kv_len_dev is the step's position, the new row included
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int split = blockIdx.x, nsplit = gridDim.x, kvh = blockIdx.y, b = blockIdx.z;
    const int G = Hq / Hkv;
    const int ntiles = (kv_len + 63) / 64;
    // left-padded batch: sequence b lives in cache rows kv_lo .. kv_len - 1; the splits divide THAT range
    const int kv_lo = kv_start ? max(min(kv_start[b], kv_len - 1), 0) : 0;
    const int t_lo = kv_lo >> 6;
    const int per = (ntiles - t_lo + nsplit - 1) / nsplit;
    const int t0 = t_lo + split * per, t1 = min(ntiles, t0 + per);
    bf16_t* Kp = K + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    bf16_t* Vp = V + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    const unsigned slab = (unsigned)kv_stride * HD * 2u;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)slab, 0x00020000);
    char* ks = smem + wave * 16384;
    char* vs = ks + 8192;
    // DMA piece i = rows 8i .. 8i+7 of the tile (1 KiB, lane-linear in LDS). K: row r keeps its 16-byte chunk c at
    // position c ^ key(r), key(r) = (r >> 1) & 7 = (lane >> 4) | ((i & 1) << 2). V: chunk c at c ^ 4 ((r >> 1) & 1) — the
    // four rows of a transposing read's block then cover all 64 banks once (attention_bf16.hip) — the same for every piece
    const int sub = lane >> 3, chunk = lane & 7, kq = lane >> 4;
    int offK[2];
#pragma unroll
    for (int par = 0; par < 2; ++par) offK[par] = sub * 128 + ((chunk ^ (kq | (par << 2))) << 4);
    const int offV = sub * 128 + ((chunk ^ (((sub >> 1) & 1) << 2)) << 4);
    auto stage = [&](int t) {
        const unsigned base = (unsigned)t * 64u * 128u;         // 64 kv rows of 128 B, K and V alike
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, LDS_AS(ks + i * 1024), 16,
                                                     offK[i & 1] + (int)(base + (unsigned)i * 1024u), 0, 0, DA_KV_AUX);
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_AS(vs + i * 1024), 16,
                                                     offV + (int)(base + (unsigned)i * 1024u), 0, 0, DA_KV_AUX);
    };
    bf16x8 qf[NKD];
    const int rp = max(kv_len - 1 - (kv_start ? kv_start[b] : 0), 0);         // fused: RoPE position of the step's token
    if (fa.raw) {
        load_q_rotated<HD>(qf, fa, b, Hq, Hkv, kvh * G + min(l31, G - 1), l31 < G, h, rp);
    } else {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const bf16_t* qp = Q + ((int64_t)b * Hq + kvh * G + min(l31, G - 1)) * q_stride;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
            qf[kd] = l31 < G ? *reinterpret_cast<const bf16x8*>(qp + kd * 16 + h * 8) : z;
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int koff[2], kkey[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int row = blk * 32 + prow;
        koff[blk] = row * 128;
        kkey[blk] = (row >> 1) & 7;
    }
    // this lane's chunk of its 16-lane group's [4 kv][16 d] block: kv row 8h + (i >> 2) of a 16-kv step, d columns
    // 16 ((lane >> 4) & 1) + 4 (i & 3) of a 32-d block; d-block 1 = chunk index + 4 = ^ 64 bytes
    int vtr;
    {
        const int i = lane & 15, r = 8 * h + (i >> 2), col = 16 * ((lane >> 4) & 1) + 4 * (i & 3);
        vtr = r * 128 + ((((col >> 3) ^ (((r >> 1) & 1) << 2)) << 4) | ((col & 7) << 1));
    }

    int t = t0 + wave;
    if (t < t1) stage(t);
    for (; t < t1; t += 4) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (fa.raw && t == ntiles - 1)           // the tile that holds the step's own row: this wave appends it
            append_new_kv<HD, true, true>(fa, Kp, Vp, ks, vs, b, Hq, Hkv, kvh, lane,
                                          kv_len - 1, rp, [](int r) { return (r >> 1) & 7; }, [](int r) { return ((r >> 1) & 1) << 2; });
        bf16x8 kf[2][NKD], vf[NDB][4];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int kd = 0; kd < NKD; ++kd)
                kf[blk][kd] = *reinterpret_cast<const bf16x8*>(ks + koff[blk] + (((kd * 2 + h) ^ kkey[blk]) << 4));
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) vf[d][c4] = v_frag_tr(vs + c4 * 16 * 128 + (vtr ^ (d << 6)), 4 * 128);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (t + 4 < t1) stage(t + 4);            // re-arm the region: its fragments are in registers now
        __builtin_amdgcn_sched_barrier(0);
        const int kv0 = t * 64;
        f32x16 s[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int kd = 0; kd < NKD; ++kd)
                s[blk] = MFMA_32x32x16(kf[blk][kd], qf[kd], kd == 0 ? zero16 : s[blk]);
        if (kv0 + 64 > kv_len || kv0 < kv_lo) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + blk * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                    s[blk][r] = (kv < kv_len && kv >= kv_lo) ? s[blk][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float ps = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s[blk][r] - m_use); ps += p[r]; }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                u32x4 w;
                w[0] = cvt_pk_d(p[tt * 8 + 0], p[tt * 8 + 1]);
                w[1] = cvt_pk_d(p[tt * 8 + 2], p[tt * 8 + 3]);
                w[2] = cvt_pk_d(p[tt * 8 + 4], p[tt * 8 + 5]);
                w[3] = cvt_pk_d(p[tt * 8 + 6], p[tt * 8 + 7]);
                pf[blk][tt] = __builtin_bit_cast(bf16x8, w);
            }
        }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    o[d] = MFMA_32x32x16(vf[d][blk * 2 + tt], pf[blk][tt], o[d]);
        }
    }
    // ---- merge the four waves (LDS), one partial per block
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (l31 < G) {
        float* pp = &red[wave][l31][0];
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) pp[d * 32 + g4 * 8 + h * 4 + e] = o[d][g4 * 4 + e];
        if (h == 0) { pp[HD] = m_run; pp[HD + 1] = l_tot; }
    }
    __syncthreads();
    for (int idx = tid; idx < G * HD; idx += 256) {
        const int g = idx / HD, d = idx % HD;
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) m = fmaxf(m, red[w][g][HD]);
        const float m_use = m == -INFINITY ? 0.f : m;
        float acc = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __builtin_amdgcn_exp2f(red[w][g][HD] - m_use);
            acc += sc * red[w][g][d];
            l += sc * red[w][g][HD + 1];
        }
        if (O_direct) {      // a single split: this IS the result (what decode_combine_kernel computes for nsplit = 1)
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            O_direct[((int64_t)b * Hq + kvh * G + g) * HD + d] = f2bf(acc * inv);
            continue;
        }
        float* pp = part + ((((int64_t)b * Hkv + kvh) * nsplit + split) * G + g) * (HD + 2);
        pp[d] = acc;
        if (d == 0) { pp[HD] = m; pp[HD + 1] = l; }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// head_dim 128 (Llama-3.1-8B): a 64-kv tile is 16 KiB of K (256-byte rows) + 16 KiB of V per wave.
// Pulling both into registers before re-arming the region (what the head_dim-64 kernel does) would need 128 VGPRs of
// fragments on top of the 64 of O, so K and V are two phases with their own DMA groups and counted waits:
//   wait K(t) [vmcnt: the 16 V pieces behind it may stay out] -> K fragments -> re-arm K with tile t+4 -> QK^T, softmax
//   wait V(t) [the 16 K pieces just issued may stay out]      -> V fragments -> re-arm V             -> PV
// 128 KiB + merge buffer per block: one block (4 waves) per CU, the same 128 KiB of loads in flight per CU.
__global__ __launch_bounds__(256, 1) void decode_attn_lds128_kernel(const bf16_t* __restrict__ Q, bf16_t* K, bf16_t* V,
                                                                    float* __restrict__ part,
                                                                    int Hq, int Hkv, int kv_stride,
                                                                    const int32_t* __restrict__ kv_len_dev, const int32_t* __restrict__ kv_start,
                                                                    bf16_t* __restrict__ O_direct, const decode_qkv_args fa, int64_t q_stride) {
    constexpr int HD = 128, NKD = HD / 16, NDB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];     // [4 waves][K 16 KiB | V 16 KiB] then red
    float (*red)[8][HD + 2] = reinterpret_cast<float (*)[8][HD + 2]>(smem + 4 * 32768);
    const int kv_len = kv_len_dev[0] + (fa.raw ? 1 : 0);            // fused: kv_len_dev is the step's position, the new row included
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int split = blockIdx.x, nsplit = gridDim.x, kvh = blockIdx.y, b = blockIdx.z;
    const int G = Hq / Hkv;
    const int ntiles = (kv_len + 63) / 64;
    // left-padded batch: sequence b lives in cache rows kv_lo .. kv_len - 1; the splits divide THAT range
    const int kv_lo = kv_start ? max(min(kv_start[b], kv_len - 1), 0) : 0;
    const int t_lo = kv_lo >> 6;
    const int per = (ntiles - t_lo + nsplit - 1) / nsplit;
    const int t0 = t_lo + split * per, t1 = min(ntiles, t0 + per);
    bf16_t* Kp = K + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    bf16_t* Vp = V + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    const unsigned slab = (unsigned)kv_stride * HD * 2u;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)slab, 0x00020000);
    char* ks = smem + wave * 32768;
    char* vs = ks + 16384;
    // K piece i = rows 4i .. 4i+3 (256 B each); row r keeps its 16-byte chunk c at position c ^ (r & 15)
    // V piece i = rows 4i .. 4i+3 too; row r keeps chunk c at c ^ 4 (r & 3): the four rows of a transposing read's block are
    // 256 B apart and take the four 64-byte bank groups (attention_bf16.hip) — the same offsets for every piece
    int offK[4];
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const int r = lane >> 4;                                    // row inside the piece
        const int c = (lane & 15) ^ ((q4 * 4 + r) & 15);
        offK[q4] = r * 256 + (c << 4);
    }
    const int offV = (lane >> 4) * 256 + (((lane & 15) ^ ((lane >> 4) << 2)) << 4);
    auto stageK = [&](int t) {
        const unsigned kbase = (unsigned)t * 64u * 256u;        // 64 kv rows of 256 B
#pragma unroll
        for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, LDS_AS(ks + i * 1024), 16,
                                                     offK[i & 3] + (int)(kbase + (unsigned)i * 1024u), 0, 0, DA_KV_AUX);
    };
    auto stageV = [&](int t) {
        const unsigned vbase = (unsigned)t * 64u * 256u;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_AS(vs + i * 1024), 16,
                                                     offV + (int)(vbase + (unsigned)i * 1024u), 0, 0, DA_KV_AUX);
    };
    bf16x8 qf[NKD];
    const int rp = max(kv_len - 1 - (kv_start ? kv_start[b] : 0), 0);         // fused: RoPE position of the step's token
    if (fa.raw) {
        load_q_rotated<HD>(qf, fa, b, Hq, Hkv, kvh * G + min(l31, G - 1), l31 < G, h, rp);
    } else {
        const bf16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        const bf16_t* qp = Q + ((int64_t)b * Hq + kvh * G + min(l31, G - 1)) * q_stride;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
            qf[kd] = l31 < G ? *reinterpret_cast<const bf16x8*>(qp + kd * 16 + h * 8) : z;
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int koff[2], kkey[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int row = blk * 32 + prow;
        koff[blk] = row * 256;
        kkey[blk] = row & 15;
    }
    int vtr;            // see decode_attn_lds_kernel; 256-byte rows, chunk key 4 (r & 3)
    {
        const int i = lane & 15, r = 8 * h + (i >> 2), col = 16 * ((lane >> 4) & 1) + 4 * (i & 3);
        vtr = r * 256 + ((((col >> 3) ^ ((r & 3) << 2)) << 4) | ((col & 7) << 1));
    }

    int t = t0 + wave;
    if (t < t1) { stageK(t); stageV(t); }
    for (; t < t1; t += 4) {
        const bool more = t + 4 < t1;
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");          // K(t) landed; the 16 V(t) pieces may still be out
        const bool own_new = fa.raw && t == ntiles - 1;            // the tile that holds the step's own row (this wave's last tile)
        if (own_new)
            append_new_kv<HD, true, false>(fa, Kp, Vp, ks, vs, b, Hq, Hkv, kvh, lane,
                                           kv_len - 1, rp, [](int r) { return r & 15; }, [](int r) { return (r & 3) << 2; });
        f32x16 s[2];
        {
            bf16x8 kf[2][NKD];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int kd = 0; kd < NKD; ++kd)
                    kf[blk][kd] = *reinterpret_cast<const bf16x8*>(ks + koff[blk] + (((kd * 2 + h) ^ kkey[blk]) << 4));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            if (more) stageK(t + 4);             // re-arm the K half: its fragments are in registers now
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int kd = 0; kd < NKD; ++kd)
                    s[blk] = MFMA_32x32x16(kf[blk][kd], qf[kd], kd == 0 ? zero16 : s[blk]);
        }
        const int kv0 = t * 64;
        if (kv0 + 64 > kv_len || kv0 < kv_lo) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + blk * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                    s[blk][r] = (kv < kv_len && kv >= kv_lo) ? s[blk][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        const float m_new = fmaxf(m_run, mx);
        const float m_use = m_new == -INFINITY ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_use);
        m_run = m_new;
        float ps = 0.f;
        bf16x8 pf[2][2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(s[blk][r] - m_use); ps += p[r]; }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                u32x4 w;
                w[0] = cvt_pk_d(p[tt * 8 + 0], p[tt * 8 + 1]);
                w[1] = cvt_pk_d(p[tt * 8 + 2], p[tt * 8 + 3]);
                w[2] = cvt_pk_d(p[tt * 8 + 4], p[tt * 8 + 5]);
                w[3] = cvt_pk_d(p[tt * 8 + 6], p[tt * 8 + 7]);
                pf[blk][tt] = __builtin_bit_cast(bf16x8, w);
            }
        }
        l_run = l_run * alpha + ps;
        // V(t): everything issued before K(t+4) has to be in; the 16 K pieces just issued may stay out
        if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (own_new)                            // (the key row's global stores sit behind vmcnt(0) too: `more` is false here)
            append_new_kv<HD, false, true>(fa, Kp, Vp, ks, vs, b, Hq, Hkv, kvh, lane,
                                           kv_len - 1, rp, [](int r) { return r & 15; }, [](int r) { return (r & 3) << 2; });
        bf16x8 vf[NDB][4];
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) vf[d][c4] = v_frag_tr(vs + c4 * 16 * 256 + (vtr ^ (d << 6)), 4 * 256);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) stageV(t + 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
                    o[d] = MFMA_32x32x16(vf[d][blk * 2 + tt], pf[blk][tt], o[d]);
        }
    }
    // ---- merge the four waves (LDS), one partial per block — as in decode_attn_lds_kernel
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    if (l31 < G) {
        float* pp = &red[wave][l31][0];
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
                for (int e = 0; e < 4; ++e) pp[d * 32 + g4 * 8 + h * 4 + e] = o[d][g4 * 4 + e];
        if (h == 0) { pp[HD] = m_run; pp[HD + 1] = l_tot; }
    }
    __syncthreads();
    for (int idx = tid; idx < G * HD; idx += 256) {
        const int g = idx / HD, d = idx % HD;
        float m = -INFINITY;
#pragma unroll
        for (int w = 0; w < 4; ++w) m = fmaxf(m, red[w][g][HD]);
        const float m_use = m == -INFINITY ? 0.f : m;
        float acc = 0.f, l = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float sc = __builtin_amdgcn_exp2f(red[w][g][HD] - m_use);
            acc += sc * red[w][g][d];
            l += sc * red[w][g][HD + 1];
        }
        if (O_direct) {      // a single split: this IS the result (what decode_combine_kernel computes for nsplit = 1)
            const float inv = l > 0.f ? 1.0f / l : 0.f;
            O_direct[((int64_t)b * Hq + kvh * G + g) * HD + d] = f2bf(acc * inv);
            continue;
        }
        float* pp = part + ((((int64_t)b * Hkv + kvh) * nsplit + split) * G + g) * (HD + 2);
        pp[d] = acc;
        if (d == 0) { pp[HD] = m; pp[HD + 1] = l; }
    }
}

// one wave per (b, q head): lanes over splits for the statistics, then over d for the accumulation
template <int HD>
__global__ __launch_bounds__(64) void decode_combine_kernel(const float* __restrict__ part, bf16_t* __restrict__ O, int Hq,
                                                            int Hkv, int nsplit) {
    const int head = blockIdx.x, b = blockIdx.y, lane = threadIdx.x;
    const int G = Hq / Hkv, kvh = head / G, g = head % G;
    const float* base = part + ((((int64_t)b * Hkv + kvh) * nsplit) * G + g) * (HD + 2);
    const int64_t sstride = (int64_t)G * (HD + 2);
    const float ms = lane < nsplit ? base[lane * sstride + HD] : -INFINITY;
    const float ls = lane < nsplit ? base[lane * sstride + HD + 1] : 0.f;
    const float m = wave_max(ms);
    const float m_use = m == -INFINITY ? 0.f : m;
    const float wgt = __builtin_amdgcn_exp2f(ms - m_use);          // 0 for empty splits / lanes >= nsplit
    const float l = wave_sum(wgt * ls);
    float acc[HD / 64];
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) acc[i] = 0.f;
#pragma unroll 8
    for (int s = 0; s < nsplit; ++s) {
        const float w = __shfl(wgt, s, 64);
#pragma unroll
        for (int i = 0; i < HD / 64; ++i) acc[i] += w * base[s * sstride + i * 64 + lane];
    }
    const float inv = l > 0.f ? 1.0f / l : 0.f;
#pragma unroll
    for (int i = 0; i < HD / 64; ++i) O[((int64_t)b * Hq + head) * HD + i * 64 + lane] = f2bf(acc[i] * inv);
}

extern "C" int64_t gar_attention_decode_workspace(int B, int Hq, int hd, int max_splits) {
    return (int64_t)B * Hq * max_splits * (hd + 2) * 4;
}

static int launch_decode(const void* q, int64_t q_stride, void* Kc, void* Vc, void* O, int B, int Hq, int Hkv, int hd, int Smax,
                         const int32_t* kv_len_dev, const int32_t* kv_start, int max_splits, void* workspace,
                         const decode_qkv_args& fa, gar_stream_t stream) {
    GAR_CHECK_ARG(workspace && max_splits > 0 && max_splits <= 64, "attention_decode: workspace / max_splits (1..64)");
    if ((int64_t)Smax * hd * 2 >= ((int64_t)1 << 31)) {       // one (sequence, kv head) slab has to fit a buffer descriptor
        gar_set_error("attention_decode: Smax %d x head_dim %d exceeds the 2 GiB per-slab range of the DMA-staged kernels", Smax, hd);
        return GAR_ERR_UNSUPPORTED;
    }
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(max_splits, Hkv, B);
    // one split per (sequence, kv head): the attention kernel normalises and writes O itself, no combine launch
    bf16_t* direct = max_splits == 1 ? (bf16_t*)O : nullptr;
    if (hd == 64) {
        constexpr int lds = 4 * 16384 + 4 * 8 * (64 + 2) * 4;
        static gar_once_per_device attr_once;
        attr_once.run([&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_lds_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
        hipLaunchKernelGGL(decode_attn_lds_kernel, grid, dim3(256), lds, s, (const bf16_t*)q, (bf16_t*)Kc,
                           (bf16_t*)Vc, (float*)workspace, Hq, Hkv, Smax, kv_len_dev, kv_start, direct, fa, q_stride);
        if (!direct)
            hipLaunchKernelGGL((decode_combine_kernel<64>), dim3(Hq, B), dim3(64), 0, s, (const float*)workspace,
                               (bf16_t*)O, Hq, Hkv, max_splits);
    } else {
        constexpr int lds = 4 * 32768 + 4 * 8 * (128 + 2) * 4;
        static gar_once_per_device attr_once;
        attr_once.run([&] {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&decode_attn_lds128_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        });
        hipLaunchKernelGGL(decode_attn_lds128_kernel, grid, dim3(256), lds, s, (const bf16_t*)q, (bf16_t*)Kc,
                           (bf16_t*)Vc, (float*)workspace, Hq, Hkv, Smax, kv_len_dev, kv_start, direct, fa, q_stride);
        if (!direct)
            hipLaunchKernelGGL((decode_combine_kernel<128>), dim3(Hq, B), dim3(64), 0, s, (const float*)workspace,
                               (bf16_t*)O, Hq, Hkv, max_splits);
    }
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_attention_decode(int dtype, const void* q, int64_t q_stride, const void* Kc, const void* Vc, void* O, int B,
                                    int Hq, int Hkv, int hd, int Smax, const int32_t* kv_len_dev, const int32_t* kv_start,
                                    int max_splits, void* workspace, gar_stream_t stream) {
    GAR_CHECK_ARG(q && Kc && Vc && O && kv_len_dev, "attention_decode: null pointer");
    GAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= 8, "attention_decode: Hq/Hkv must be <= 8");
    GAR_CHECK_ARG(Smax % 64 == 0, "attention_decode: Smax must be a multiple of 64");
    GAR_CHECK_ARG(hd == 64 || hd == 128, "attention_decode: head_dim %d not built (64, 128)", hd);
    if (q_stride <= 0) q_stride = hd;
    GAR_CHECK_ARG(q_stride % hd == 0 && ((uintptr_t)q & 15) == 0, "attention_decode: q_stride must be a multiple of head_dim, q 16-byte aligned");
    if (dtype == GAR_F32)   // parity mode: the prefill kernel with q_len = 1 (exact-f32 MFMA), no split; (b, head) rows q_stride apart
        return gar_attention_vrow(dtype, q, Kc, Vc, O, B, Hq, Hkv, hd, 1, (int)(q_stride / hd), 0, Smax, 0, kv_len_dev, kv_start, 0,
                                  stream);
    GAR_CHECK_ARG(dtype == GAR_BF16, "attention_decode: bad dtype");
    const decode_qkv_args none = {nullptr, nullptr, nullptr, 0.f, 0};
    // (the plain form only reads the caches)
    return launch_decode(q, q_stride, const_cast<void*>(Kc), const_cast<void*>(Vc), O, B, Hq, Hkv, hd, Smax, kv_len_dev, kv_start,
                         max_splits, workspace, none, stream);
}

// gar_llm_qkv_post (S = 1) + gar_attention_decode in ONE launch (bf16): see decode_qkv_args. GAR_ERR_UNSUPPORTED (nothing
// launched) in parity mode: the caller keeps the two calls.
extern "C" int gar_attention_decode_qkv(int dtype, const void* qkv, const float* cos, const float* sin, void* Kc, void* Vc,
                                        void* O, int B, int Hq, int Hkv, int hd, int Smax, const int32_t* pos_dev,
                                        const int32_t* left_pad, float q_scale, int qk_strip_order, int max_splits,
                                        void* workspace, gar_stream_t stream) {
    GAR_CHECK_ARG(qkv && cos && sin && Kc && Vc && O && pos_dev, "attention_decode_qkv: null pointer");
    GAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0 && Hq / Hkv <= 8, "attention_decode_qkv: Hq/Hkv must be <= 8");
    GAR_CHECK_ARG(Smax % 64 == 0, "attention_decode_qkv: Smax must be a multiple of 64");
    GAR_CHECK_ARG(hd == 64 || hd == 128, "attention_decode_qkv: head_dim %d not built (64, 128)", hd);
    GAR_CHECK_ARG(((uintptr_t)qkv & 15) == 0, "attention_decode_qkv: qkv must be 16-byte aligned");
    if (dtype != GAR_BF16) {
        gar_set_error("attention_decode_qkv: bf16 only (parity mode runs gar_llm_qkv_post + gar_attention_decode)");
        return GAR_ERR_UNSUPPORTED;
    }
    const decode_qkv_args fa = {(const bf16_t*)qkv, cos, sin, q_scale, qk_strip_order ? 1 : 0};
    return launch_decode(nullptr, hd, Kc, Vc, O, B, Hq, Hkv, hd, Smax, pos_dev, left_pad, max_splits, workspace, fa, stream);
}
