// Flash-style attention for gfx950 (ViT tile attention, Llama causal GQA prefill, and — with q_len = 1 and the kv
// length read from device memory — the decode step).
//
// Layouts chosen so that NO cross-lane traffic is needed between QK^T, softmax and PV:
//   * scores are computed TRANSPOSED, S^T = K Q^T with v_mfma_f32_32x32x16_bf16 (A = K rows, B = Q rows): lane
//     (q = lane&31, h = lane>>5) then holds 16 kv values of ONE query per 32-kv block, so the online-softmax row
//     statistics are per-lane (+ one exchange between the two half-waves);
//   * K rows are fed in a permuted order (bits 2<->3 of the row index swapped) which makes the accumulator
//     registers 8t..8t+7 of that lane exactly the kv = 16t + 8h + (0..7) slice the PV MFMA wants as its B operand:
//     P goes exp2 -> bf16 pack -> MFMA operand without leaving the lane;
//   * gar_attention takes V transposed (Vt [hd][kv]; the f32 ViT path) so O^T = Vt P^T takes Vt rows as plain
//     16-byte A fragments; O^T lands as lane (q, h) <- 16 d values: the per-query rescale is per-lane too.
// K / Vt tiles are staged in LDS (XOR-swizzled 16-byte chunks, ds_read_b128 conflict-free for these access sets),
// register-prefetched one tile ahead (issue loads -> compute current tile -> write next tile -> barrier).
//
// f32 variant (parity mode): same structure on v_mfma_f32_32x32x2_f32 (exact f32 fma chains), padded LDS tiles.
#include "common.h"

// attention_bf16.hip: DMA-staged bf16 kernel (default for bf16); false -> use the register-staged kernel below
bool gar_attn_bf16_v2_try(const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv, int hd, int q_len,
                          int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev, int vrow,
                          const int32_t* kv_start, int kv_prefix, hipStream_t s, int o_rows = 0);
#ifdef GAR_ATTN_V4_VARIANT   /* diagnostic build only (tools/attn_v4/: the round-5 4-wave persistent kernel, slower than v2) */
// query rows q_row0 .. q_row0 + q_len - 1 of sequences of q_total rows; false -> not built for this problem
bool gar_attn_bf16_v4_try(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv, int hd, int q_row0,
                          int q_len, int q_total, int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev,
                          const int32_t* kv_start, int kv_prefix, hipStream_t s);
#endif

typedef float f32v2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int cvt_pk_bf16(float lo, float hi) { return pack_bf2(lo, hi); }

// swizzle key of a row for a tile whose rows are RS bytes: makes 16 rows read at one logical chunk hit 16 distinct
// 16-byte slots of the 256-byte LDS bank row ("attention LDS image").
#define RESCALE_THR 6.0f   // log2 domain

template <int RS> __device__ __forceinline__ int swz_key(int row) { return RS == 128 ? ((row >> 1) & 7) : (row & 15); }

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256) void attn_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                        const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O, int Hq,
                                                        int Hkv, int q_len, int q_pad, int kv_len_arg, int kv_stride,
                                                        const int32_t* __restrict__ kv_len_dev,
                                                        const int32_t* __restrict__ kv_start) {
    constexpr int KRS = HD * 2;                 // K tile row bytes
    constexpr int KT = 64 * KRS;                // K tile bytes   [64 kv][HD]
    constexpr int VT = HD * 128;                // Vt tile bytes  [HD][64 kv]
    constexpr int NKD = HD / 16;                // QK^T k-steps
    constexpr int NDB = HD / 32;                // O^T row blocks
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][K | Vt]
    const int kv_len = kv_len_dev ? kv_len_dev[0] : kv_len_arg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: tile-skip / mask branches stay uniform
    const int l31 = lane & 31, h = lane >> 5;
    const int qb = gridDim.x - 1 - blockIdx.x;  // heavy (late) causal blocks first
    const int head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / (Hq / Hkv);
    const int q0 = qb * 128 + wave * 32;        // this wave's first query
    const int coff = kv_len - q_len;            // causal: kv <= q + coff
    const bf16_t* Qp = Q + (((int64_t)b * Hq + head) * q_pad) * HD;
    const bf16_t* Kp = K + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    const bf16_t* Vp = Vt + (((int64_t)b * Hkv + kvh) * HD) * (int64_t)kv_stride;

    // Q fragments (B operand): Q[q0 + l31][16 kd + 8h .. +8]
    bf16x8 qf[NKD];
    {
        const int qrow = min(q0 + l31, q_len - 1);        // rows >= q_len are never stored: a pointer into a larger Q stays in bounds
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

    // kv range for this block (kv_start: left-padded batch, see attention_bf16.hip)
    const int kv_lo = kv_start ? max(min(kv_start[b], kv_len - 1), 0) : 0;
    const int t_lo = kv_lo >> 6;
    int kv_end = kv_len;
    if (CAUSAL) kv_end = min(kv_len, max(qb * 128 + 127 + coff, kv_lo) + 1);
    const int ntiles = (kv_end + 63) / 64;

    // staging map: K tile = 64 rows x (KRS/16) chunks; Vt tile = HD rows x 8 chunks; 256 threads, 16 B each
    constexpr int KCH = KRS / 16;
    constexpr int KPASS = 64 * KCH / 256;
    constexpr int VPASS = HD * 8 / 256;
    uint4 kreg[KPASS], vreg[VPASS];
    auto load_tile = [&](int t) {
        const int kv0 = t * 64;
#pragma unroll
        for (int p = 0; p < KPASS; ++p) {
            const int idx = p * 256 + tid;
            const int row = idx / KCH, c = idx % KCH;
            const int kvr = min(kv0 + row, kv_stride - 1);
            kreg[p] = *reinterpret_cast<const uint4*>(Kp + (int64_t)kvr * HD + c * 8);
        }
#pragma unroll
        for (int p = 0; p < VPASS; ++p) {
            const int idx = p * 256 + tid;
            const int row = idx / 8, c = idx % 8;
            vreg[p] = *reinterpret_cast<const uint4*>(Vp + (int64_t)row * kv_stride + kv0 + c * 8);
        }
    };
    auto write_tile = [&](int buf) {
        char* ks = smem + buf * (KT + VT);
        char* vs = ks + KT;
#pragma unroll
        for (int p = 0; p < KPASS; ++p) {
            const int idx = p * 256 + tid;
            const int row = idx / KCH, c = idx % KCH;
            *reinterpret_cast<uint4*>(ks + row * KRS + ((c ^ swz_key<KRS>(row)) << 4)) = kreg[p];
        }
#pragma unroll
        for (int p = 0; p < VPASS; ++p) {
            const int idx = p * 256 + tid;
            const int row = idx / 8, c = idx % 8;
            *reinterpret_cast<uint4*>(vs + row * 128 + ((c ^ swz_key<128>(row)) << 4)) = vreg[p];
        }
    };

    if (ntiles > t_lo) {
        load_tile(t_lo);
        write_tile(t_lo & 1);
    }
    __syncthreads();
    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2<->3 swapped
    const bool wave_active = q0 < q_len;
    for (int t = t_lo; t < ntiles; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntiles) load_tile(t + 1);
        const int kv0 = t * 64;
        // wave-uniform skip: tile entirely above this wave's causal diagonal
        const bool skip = !wave_active || (CAUSAL && kv0 > max(q0 + 31 + coff, kv_lo));
        if (!skip) {
            const char* ks = smem + buf * (KT + VT);
            const char* vs = ks + KT;
            f32x16 s[2];
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
                for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
                const int row = blk * 32 + prow;
                const int key = swz_key<KRS>(row);
#pragma unroll
                for (int kd = 0; kd < NKD; ++kd) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + row * KRS + (((kd * 2 + h) ^ key) << 4));
                    s[blk] = MFMA_32x32x16(kf, qf[kd], s[blk]);
                }
            }
            // masking (only on boundary tiles). register r of block blk <-> kv = kv0 + 32 blk + 16 (r>>3) + 8 h + (r&7)
            const int qi = q0 + l31;
            const bool need_mask = (kv0 + 64 > kv_len) || kv0 < kv_lo || (CAUSAL && kv0 + 63 > q0 + coff);
            if (need_mask) {
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
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // deferred rescale: keep the old reference max while no row's max grew by more than 2^RESCALE_THR
            // (P <= 2^THR is harmless in bf16/f32); the O / l rescale then runs only on the few tiles where it matters.
            if (!__all(mx - m_run <= RESCALE_THR)) {           // NaN (-inf - -inf) also lands here
                const float m_new = fmaxf(m_run, mx);
                const float m_nu = m_new == -INFINITY ? 0.f : m_new;
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_nu);      // m_run = -inf -> 0
                m_run = m_new;
                l_run *= alpha;
#pragma unroll
                for (int d = 0; d < NDB; ++d)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
            }
            const float m_use = m_run == -INFINITY ? 0.f : m_run;
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
                    w[0] = cvt_pk_bf16(p[tt * 8 + 0], p[tt * 8 + 1]);
                    w[1] = cvt_pk_bf16(p[tt * 8 + 2], p[tt * 8 + 3]);
                    w[2] = cvt_pk_bf16(p[tt * 8 + 4], p[tt * 8 + 5]);
                    w[3] = cvt_pk_bf16(p[tt * 8 + 6], p[tt * 8 + 7]);
                    pf[blk][tt] = __builtin_bit_cast(bf16x8, w);
                }
            }
            l_run += ps;
#pragma unroll
            for (int d = 0; d < NDB; ++d) {
                const int row = d * 32 + l31;
                const int key = swz_key<128>(row);
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const int c = (blk * 2 + tt) * 2 + h;
                        const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs + row * 128 + ((c ^ key) << 4));
                        o[d] = MFMA_32x32x16(vf, pf[blk][tt], o[d]);
                    }
            }
        }
        if (t + 1 < ntiles) write_tile(buf ^ 1);
        __syncthreads();
    }
    // epilogue: O[b*q_len + q][head*HD + d], d = 32 db + (r&3) + 8 (r>>2) + 4 h
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int qi = q0 + l31;
    if (qi < q_len) {
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        bf16_t* op = O + ((int64_t)b * q_len + qi) * ((int64_t)Hq * HD) + head * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {o[d][g * 4 + 0] * inv, o[d][g * 4 + 1] * inv, o[d][g * 4 + 2] * inv, o[d][g * 4 + 3] * inv};
                st4(op + d * 32 + g * 8 + h * 4, v);
            }
    }
}

// ===============================================================================================================
// f32 (parity mode) — v_mfma_f32_32x32x2_f32; lane (i = lane&31, k = lane>>5) supplies A[i][k] / B[k][i]
// ===============================================================================================================
// VROW: V row-major [kv][HD] (the KV cache's layout); the tile is transposed on its way into LDS.
template <int HD, bool CAUSAL, bool VROW = false>
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                       const float* __restrict__ Vt, float* __restrict__ O, int Hq,
                                                       int Hkv, int q_len, int q_pad, int kv_len_arg, int kv_stride,
                                                       const int32_t* __restrict__ kv_len_dev,
                                                       const int32_t* __restrict__ kv_start) {
    constexpr int KLD = HD + 1;
    constexpr int VLD = 65;
    constexpr int NDB = HD / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* ks = reinterpret_cast<float*>(smem);           // [64][KLD]
    float* vs = ks + 64 * KLD;                            // [HD][VLD]
    const int kv_len = kv_len_dev ? kv_len_dev[0] : kv_len_arg;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int qb = gridDim.x - 1 - blockIdx.x;
    const int head = blockIdx.y, b = blockIdx.z;
    const int kvh = head / (Hq / Hkv);
    const int q0 = qb * 128 + wave * 32;
    const int coff = kv_len - q_len;
    const float* Qp = Q + (((int64_t)b * Hq + head) * q_pad) * HD;
    const float* Kp = K + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    const float* Vp = Vt + (((int64_t)b * Hkv + kvh) * HD) * (int64_t)kv_stride;       // same slab offset in either layout
    float qf[HD / 2];
    {
        const int qrow = min(q0 + l31, q_len - 1);        // rows >= q_len are never stored: a pointer into a larger Q stays in bounds
#pragma unroll
        for (int k = 0; k < HD / 2; ++k) qf[k] = Qp[(int64_t)qrow * HD + 2 * k + h];
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int kv_lo = kv_start ? max(min(kv_start[b], kv_len - 1), 0) : 0;       // left-padded batch, see attention_bf16.hip
    const int t_lo = kv_lo >> 6;
    int kv_end = kv_len;
    if (CAUSAL) kv_end = min(kv_len, max(qb * 128 + 127 + coff, kv_lo) + 1);
    const int ntiles = (kv_end + 63) / 64;
    const bool wave_active = q0 < q_len;
    for (int t = t_lo; t < ntiles; ++t) {
        const int kv0 = t * 64;
        __syncthreads();
        for (int idx = tid; idx < 64 * (HD / 4); idx += 256) {
            const int row = idx / (HD / 4), c4 = (idx % (HD / 4)) * 4;
            const int kvr = min(kv0 + row, kv_stride - 1);
            const float4 v = *reinterpret_cast<const float4*>(Kp + (int64_t)kvr * HD + c4);
            float* d = ks + row * KLD + c4;
            d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
        }
        if (VROW) {
            for (int idx = tid; idx < 64 * (HD / 4); idx += 256) {
                const int row = idx / (HD / 4), c4 = (idx % (HD / 4)) * 4;
                const int kvr = min(kv0 + row, kv_stride - 1);
                const float4 v = *reinterpret_cast<const float4*>(Vp + (int64_t)kvr * HD + c4);
                float* d = vs + c4 * VLD + row;
                d[0] = v.x; d[VLD] = v.y; d[2 * VLD] = v.z; d[3 * VLD] = v.w;
            }
        } else {
            for (int idx = tid; idx < HD * 16; idx += 256) {
                const int row = idx / 16, c4 = (idx % 16) * 4;
                const float4 v = *reinterpret_cast<const float4*>(Vp + (int64_t)row * kv_stride + kv0 + c4);
                float* d = vs + row * VLD + c4;
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        }
        __syncthreads();
        const bool skip = !wave_active || (CAUSAL && kv0 > max(q0 + 31 + coff, kv_lo));
        if (skip) continue;
        f32x16 s[2];
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[blk][r] = 0.f;
            const float* kr = ks + (blk * 32 + l31) * KLD + h;
#pragma unroll
            for (int k = 0; k < HD / 2; ++k)
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x2f32(kr[2 * k], qf[k], s[blk], 0, 0, 0);
        }
        // register r of block blk <-> kv = kv0 + 32 blk + (r&3) + 8 (r>>2) + 4 h
        const int qi = q0 + l31;
        const bool need_mask = (kv0 + 64 > kv_len) || kv0 < kv_lo || (CAUSAL && kv0 + 63 > q0 + coff);
        if (need_mask) {
            const int lim = CAUSAL ? min(kv_len - 1, max(qi + coff, kv_lo)) : kv_len - 1;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + blk * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    s[blk][r] = (kv <= lim && kv >= kv_lo) ? s[blk][r] : -INFINITY;
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
        const float alpha = exp2f(m_run - m_use);
        m_run = m_new;
        float ps = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[blk][r] = exp2f(s[blk][r] - m_use); ps += s[blk][r]; }
        l_run = l_run * alpha + ps;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[d][r] *= alpha;
        // O^T += Vt P^T : k-step (blk, r): slot k = h <-> kv = 32 blk + (r&3) + 8 (r>>2) + 4 h
#pragma unroll
        for (int d = 0; d < NDB; ++d) {
            const float* vr = vs + (d * 32 + l31) * VLD + 4 * h;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[blk * 32 + (r & 3) + 8 * (r >> 2)], s[blk][r], o[d],
                                                                0, 0, 0);
        }
    }
    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const int qi = q0 + l31;
    if (qi < q_len) {
        const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
        float* op = O + ((int64_t)b * q_len + qi) * ((int64_t)Hq * HD) + head * HD;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v[4] = {o[d][g * 4 + 0] * inv, o[d][g * 4 + 1] * inv, o[d][g * 4 + 2] * inv, o[d][g * 4 + 3] * inv};
                st4(op + d * 32 + g * 8 + h * 4, v);
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <int HD>
static int launch_attn(int dtype, const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv,
                       int q_len, int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev,
                       const int32_t* kv_start, hipStream_t s, bool vrow = false) {
    dim3 grid((q_len + 127) / 128, Hq, B), block(256);
    if (dtype == GAR_F32 && vrow) {
        const int lds = (64 * (HD + 1) + HD * 65) * 4;
        if (causal)
            hipLaunchKernelGGL((attn_f32_kernel<HD, true, true>), grid, block, lds, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start);
        else
            hipLaunchKernelGGL((attn_f32_kernel<HD, false, true>), grid, block, lds, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start);
    } else if (dtype == GAR_BF16) {
        const int lds = 2 * (64 * HD * 2 + HD * 128);
        if (causal)
            hipLaunchKernelGGL((attn_bf16_kernel<HD, true>), grid, block, lds, s, (const bf16_t*)Q, (const bf16_t*)K,
                               (const bf16_t*)Vt, (bf16_t*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start);
        else
            hipLaunchKernelGGL((attn_bf16_kernel<HD, false>), grid, block, lds, s, (const bf16_t*)Q, (const bf16_t*)K,
                               (const bf16_t*)Vt, (bf16_t*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start);
    } else {
        const int lds = (64 * (HD + 1) + HD * 65) * 4;
        if (causal)
            hipLaunchKernelGGL((attn_f32_kernel<HD, true>), grid, block, lds, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start);
        else
            hipLaunchKernelGGL((attn_f32_kernel<HD, false>), grid, block, lds, s, (const float*)Q, (const float*)K,
                               (const float*)Vt, (float*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev, kv_start);
    }
    return GAR_OK;
}

extern "C" int gar_attention(int dtype, const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv,
                             int hd, int q_len, int q_pad, int kv_len, int kv_stride, int causal,
                             const int32_t* kv_len_dev, const int32_t* kv_start, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "attention: bad dtype");
    GAR_CHECK_ARG(Q && K && Vt && O, "attention: null pointer");
    GAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention: bad heads %d/%d", Hq, Hkv);
    GAR_CHECK_ARG(q_len > 0 && q_pad >= q_len && kv_stride % 64 == 0, "attention: bad lengths");
    GAR_CHECK_ARG(kv_len_dev || (kv_len > 0 && kv_len <= kv_stride && (!causal || kv_len >= q_len)),
                  "attention: kv_len %d out of range (stride %d, q_len %d)", kv_len, kv_stride, q_len);
    GAR_CHECK_ARG(hd == 64 || hd == 128 || (hd == 96 && dtype == GAR_BF16),
                  "attention: head_dim %d not built (64, 128; 96 in bf16 only)", hd);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16 &&
        gar_attn_bf16_v2_try(Q, K, Vt, O, B, Hq, Hkv, hd, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev, 0, kv_start, 0, s)) {
        GAR_CHECK_LAUNCH();
        return GAR_OK;
    }
    if (hd == 64) launch_attn<64>(dtype, Q, K, Vt, O, B, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev, kv_start, s);
    else launch_attn<128>(dtype, Q, K, Vt, O, B, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev, kv_start, s);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}


// Same attention with V row-major [B, Hkv, kv_stride, hd] — the layout K has, the one the fused qkv GEMM epilogues
// write (gar_gemm_params.qkv_v) and the one the Llama KV cache keeps — read through the transposing LDS load of gfx950:
// no transpose pass over V. bf16: head_dim 64 / 96 / 128; f32 (parity mode): head_dim 64 / 128, kv_prefix = 0, the tile is
// transposed while it is staged. GAR_ERR_UNSUPPORTED (nothing launched) otherwise.
extern "C" int gar_attention_vrow(int dtype, const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv,
                                  int hd, int q_len, int q_pad, int kv_len, int kv_stride, int causal,
                                  const int32_t* kv_len_dev, const int32_t* kv_start, int kv_prefix, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "attention_vrow: bad dtype");
    GAR_CHECK_ARG(Q && K && V && O, "attention_vrow: null pointer");
    GAR_CHECK_ARG(B > 0 && Hq > 0 && Hkv > 0 && Hq % Hkv == 0, "attention_vrow: bad heads %d/%d", Hq, Hkv);
    GAR_CHECK_ARG(q_len > 0 && q_pad >= q_len && kv_stride % 64 == 0, "attention_vrow: bad lengths");
    GAR_CHECK_ARG(kv_len_dev || (kv_len > 0 && kv_len <= kv_stride && (!causal || kv_len >= q_len)),
                  "attention_vrow: kv_len %d out of range (stride %d, q_len %d)", kv_len, kv_stride, q_len);
    GAR_CHECK_ARG(kv_prefix == 0 || (kv_prefix == 1 && !causal && !kv_len_dev && kv_len > 1 && dtype == GAR_BF16),
                  "attention_vrow: kv_prefix is 0 or 1 (bf16, non-causal, kv_len > 1)");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_F32 && (hd == 64 || hd == 128)) {
        if (hd == 64) launch_attn<64>(dtype, Q, K, V, O, B, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev, kv_start, s, true);
        else launch_attn<128>(dtype, Q, K, V, O, B, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev, kv_start, s, true);
        GAR_CHECK_LAUNCH();
        return GAR_OK;
    }
#ifdef GAR_ATTN_V4_VARIANT
    if (dtype == GAR_BF16 && hd == 64 && q_len >= 256 && !kv_len_dev) {
        // v4 (tools/attn_v4/attention_v4.hip) walks 256-row Q blocks. A non-causal sequence whose length is a few rows more than a multiple of
        // 256 — the ViT tile: 1 cls + 1024 patch tokens — gives those first rows to v2 (one q-block with `head` live rows costs
        // what a ninth of the tile costs) and whole blocks to v4, instead of a fifth 256-row block with one live row.
        const int head = (!causal && (q_len % 256) <= 32) ? q_len % 256 : 0;
        if (gar_attn_bf16_v4_try(Q, K, V, O, B, Hq, Hkv, hd, head, q_len - head, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev,
                                 kv_start, kv_prefix, s)) {
            if (head && !gar_attn_bf16_v2_try(Q, K, V, O, B, Hq, Hkv, hd, head, q_pad, kv_len, kv_stride, causal, kv_len_dev, 1, kv_start,
                                              kv_prefix, s, q_len)) {
                gar_set_error("attention_vrow: the head rows of a v4 launch were refused by v2");
                return GAR_ERR_UNSUPPORTED;
            }
            GAR_CHECK_LAUNCH();
            return GAR_OK;
        }
    }
#endif
    if (dtype != GAR_BF16 || (hd != 64 && hd != 96 && hd != 128) ||
        !gar_attn_bf16_v2_try(Q, K, V, O, B, Hq, Hkv, hd, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev, 1, kv_start,
                              kv_prefix, s)) {
        gar_set_error("attention_vrow: built for bf16 head_dim 64 / 96 / 128 (kv slab < 2 GiB) and f32 head_dim 64 / 128 "
                      "(dtype %d, head_dim %d)", dtype, hd);
        return GAR_ERR_UNSUPPORTED;
    }
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
