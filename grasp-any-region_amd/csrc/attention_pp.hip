// bf16 flash attention, head_dim 64, 8-wave "ping-pong" (prefill / ViT): the lane-local S^T -> softmax -> P -> O^T scheme
// of attention_bf16.hip, with the two waves that share a SIMD held in OPPOSITE phases by workgroup barriers, the way the
// tile GEMM (gemm_pp.hip) pairs its wave rows:
//
//   workgroup = 512 threads = 256 query rows; waves 0-3 (group 0) and 4-7 (group 1) sit pairwise on the four SIMDs.
//   A wave alternates two segments per 64-kv tile t, each closed by an s_barrier:
//     V_t   softmax of tile t's scores (mask, row max, deferred rescale, exp2, row sum, bf16 pack -> P_t), the DMA issue
//           of a later tile, and the ds_reads of K(t+1) / Vt(t) fragments                    (VALU / LDS / VMEM, no MFMA)
//     M_t   8 PV MFMAs of tile t + 8 QK^T MFMAs of tile t+1, back to back                    (matrix pipe only)
//   Group 1 runs the same program one barrier interval late, so in every interval each SIMD has one wave in M and one in
//   V: 16 x 32 = 512 MFMA cycles beside ~130 VALU + 32 v_exp of the partner. (v2 leaves the overlap to chance between
//   3-4 unsynchronised waves per SIMD: the matrix pipes were 42 % busy — profiles/r2_pmc_attention.json.)
//
//   K / Vt tiles: `buffer_load_dwordx4 ... lds` into rings of NS = 4 tiles each, every wave one 1-KiB piece of K and of Vt
//   per tile, issued LEAD = 3 tiles ahead in the wave's V segment and waited for with a counted vmcnt one tile before
//   the first read; the interval barriers publish the pieces and fence the slot reuse:
//     interval I = 2t     : group 0 in V_t (reads K(t+1), Vt(t) at its end), group 1 in M_{t-1}
//     interval I = 2t + 1 : group 0 in M_t,                                  group 1 in V_t
//     K(j) is issued at V_{j-1-LEAD}, waited for at V_{j-2} (by every wave, one barrier before group 0 reads it at the end
//     of V_{j-1}), last read by group 1 at I = 2j - 1; its slot is re-armed by K(j + NS) at group 0's V_j (I = 2j).
#include <stdlib.h>

#include "common.h"

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define PPA_RESCALE_THR 6.0f   // log2 domain: P <= 2^6 between rescales

typedef __bf16 ppa_bf16v2_t __attribute__((ext_vector_type(2)));
typedef float ppa_f32v2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int ppa_cvt_pk(float lo, float hi) {
    ppa_f32v2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, ppa_bf16v2_t));
}

template <bool CAUSAL>
__global__ __launch_bounds__(512, 2) void attn_pp_bf16_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                              int Hq, int Hkv, int q_len, int q_pad, int kv_len_arg,
                                                              int kv_stride, const int32_t* __restrict__ kv_len_dev) {
    constexpr int HD = 64, KT = 64 * 128, VT = HD * 128, NKD = 4, NDB = 2;
    constexpr int NS = 4, LEAD = 3;
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NS] K tiles | [NS] Vt tiles
    char* const kring = smem;
    char* const vring = smem + NS * KT;
    const int kv_len = kv_len_dev ? kv_len_dev[0] : kv_len_arg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int grp = wave >> 2;
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware 1-D grid (see attention_bf16.hip): every XCD gets one contiguous chunk of the (batch, head, q-block) list
    const int nqb = (q_len + 255) >> 8;
    int wk;
    {
        const int total = gridDim.x, L = blockIdx.x;
        const int per = total >> 3, rem = total & 7, xcd = L & 7, slot = L >> 3;
        wk = (xcd < rem ? xcd * (per + 1) : rem * (per + 1) + (xcd - rem) * per) + slot;
    }
    const int qb = nqb - 1 - (wk % nqb);        // heavy (late) causal blocks first
    const int head = (wk / nqb) % Hq, b = wk / (nqb * Hq);
    const int kvh = head / (Hq / Hkv);
    const int q0 = qb * 256 + wave * 32;        // this wave's first query
    const int coff = kv_len - q_len;            // causal: kv <= q + coff
    const bf16_t* Qp = Q + (((int64_t)b * Hq + head) * q_pad) * HD;
    const bf16_t* Kp = K + (((int64_t)b * Hkv + kvh) * (int64_t)kv_stride) * HD;
    const bf16_t* Vp = Vt + (((int64_t)b * Hkv + kvh) * HD) * (int64_t)kv_stride;
    const unsigned slab = (unsigned)kv_stride * HD * 2u;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)slab, 0x00020000);

    // one 1-KiB piece (8 rows x 128 B) of each tile per wave; LDS image lane-linear, swizzle chunk ^= (row >> 1) & 7 applied
    // on the source side
    const int prow_dma = wave * 8 + (lane >> 3);
    const int voffK = prow_dma * 128 + (((lane & 7) ^ ((prow_dma >> 1) & 7)) << 4);
    const int voffV = (int)((unsigned)prow_dma * (unsigned)kv_stride * 2u) + (((lane & 7) ^ ((prow_dma >> 1) & 7)) << 4);
    auto dma_k = [&](int t) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, LDS_AS(kring + (t & (NS - 1)) * KT + wave * 1024), 16,
                                                 voffK + (int)((unsigned)t * 64u * 128u), 0, 0, 0);
    };
    auto dma_v = [&](int t) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_AS(vring + (t & (NS - 1)) * VT + wave * 1024), 16,
                                                 voffV + (int)((unsigned)t * 128u), 0, 0, 0);
    };

    int kv_end = kv_len;
    if (CAUSAL) kv_end = min(kv_len, qb * 256 + 255 + coff + 1);
    const int ntiles = (kv_end + 63) / 64;                       // tiles of this workgroup (>= 1)
    const bool wave_active = q0 < q_len;
    int wtiles = wave_active ? ntiles : 0;                       // tiles this wave computes
    if (CAUSAL && wave_active) wtiles = min(ntiles, (min(kv_len, q0 + 31 + coff + 1) + 63) / 64);

    // Q fragments first (in-order vmcnt: waiting for them must not drain the tile prefetches behind them)
    bf16x8 qf[NKD];
    {
        const int qrow = min(q0 + l31, q_pad - 1);
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
            qf[kd] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)qrow * HD + kd * 16 + h * 8);
    }
    // prologue DMAs, in the order the steady state would have issued them: K(0), then the pairs {K(j+1), Vt(j)} for
    // j = 0 .. LEAD - 1 (V_t issues {K(t+1+LEAD), Vt(t+LEAD)}); tiles past the end are fetched all the same — nobody reads
    // them — so that the counted waits stay uniform
    dma_k(0);
#pragma unroll
    for (int j = 0; j < LEAD; ++j) { dma_k(j + 1); dma_v(j); }

    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2<->3 swapped (see attention.hip)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int koff[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int row = blk * 32 + prow;
        koff[blk] = row * 128;            // + ((chunk ^ key) << 4), key = (row >> 1) & 7
    }
    const int kkey0 = ((0 * 32 + prow) >> 1) & 7, kkey1 = ((1 * 32 + prow) >> 1) & 7;
    int voff_r[NDB], vkey[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d) {
        const int row = d * 32 + l31;
        voff_r[d] = row * 128;
        vkey[d] = (row >> 1) & 7;
    }

    bf16x8 kf[2][NKD], vf[NDB][4];
    auto read_k = [&](int t) {            // K(t) fragments -> registers
        const char* ks = kring + (t & (NS - 1)) * KT;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) {
            kf[0][kd] = *reinterpret_cast<const bf16x8*>(ks + koff[0] + (((kd * 2 + h) ^ kkey0) << 4));
            kf[1][kd] = *reinterpret_cast<const bf16x8*>(ks + koff[1] + (((kd * 2 + h) ^ kkey1) << 4));
        }
    };
    auto read_v = [&](int t) {            // Vt(t) fragments -> registers
        const char* vs = vring + (t & (NS - 1)) * VT;
#pragma unroll
        for (int d = 0; d < NDB; ++d)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4)
                vf[d][c4] = *reinterpret_cast<const bf16x8*>(vs + voff_r[d] + (((c4 * 2 + h) ^ vkey[d]) << 4));
    };
    f32x16 s[2];
    auto qk = [&]() {                     // S^T of the tile whose K fragments are in kf
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) {
            s[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[0][kd], qf[kd], kd == 0 ? zero16 : s[0], 0, 0, 0);
            s[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[1][kd], qf[kd], kd == 0 ? zero16 : s[1], 0, 0, 0);
        }
    };

    // ---- prologue: K(0) -> S_0. Outstanding per wave: K(0) + LEAD pairs; K(0) and the pair {K(1), Vt(0)} must land
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LEAD - 1)) : "memory");
    __builtin_amdgcn_s_barrier();                                   // K(0), K(1), Vt(0) published
    if (wtiles > 0) {
        read_k(0);
        qk();
    }
    __builtin_amdgcn_s_barrier();
    if (grp == 1) __builtin_amdgcn_s_barrier();                      // group 1 runs one interval late

    bf16x8 pf[2][2];
    for (int t = 0; t < ntiles; ++t) {
        // ================= V_t
        // own pieces of {K(t+2), Vt(t+1)} (issued at V_{t+1-LEAD}): all but the newest LEAD - 2 pairs have landed; the
        // barrier closing this segment publishes them one full tile before their first reader
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * (LEAD - 2)) : "memory");
        dma_k(t + 1 + LEAD);                  // slot of K(t+1+LEAD-NS) = K(t): last read at I = 2t - 1
        dma_v(t + LEAD);                      // slot of Vt(t-1): last read at I = 2t - 1
        const bool active = t < wtiles;       // wave-uniform
        const bool more = t + 1 < wtiles;
        if (active) {
            const int kv0 = t * 64;
            if ((kv0 + 64 > kv_len) || (CAUSAL && kv0 + 63 > q0 + coff)) {      // wave-uniform: tile crosses an edge
                const int qi = q0 + l31;
                const int lim = CAUSAL ? min(kv_len - 1, qi + coff) : kv_len - 1;
#pragma unroll
                for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int kv = kv0 + blk * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                        s[blk][r] = kv <= lim ? s[blk][r] : -INFINITY;
                    }
            }
            float mx0 = -INFINITY, mx1 = -INFINITY, mx2 = -INFINITY, mx3 = -INFINITY;     // four chains, then a tree
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                mx0 = fmaxf(mx0, s[0][r]); mx1 = fmaxf(mx1, s[0][r + 1]);
                mx2 = fmaxf(mx2, s[1][r]); mx3 = fmaxf(mx3, s[1][r + 1]);
            }
            float mx = fmaxf(fmaxf(mx0, mx1), fmaxf(mx2, mx3));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            if (!__all(mx - m_run <= PPA_RESCALE_THR)) {           // NaN (-inf - -inf) also lands here
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
            const float m_use = m_run == -INFINITY ? 0.f : m_run;
            float ps0 = 0.f, ps1 = 0.f, ps2 = 0.f, ps3 = 0.f;
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                float p[16];
#pragma unroll
                for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[blk][r] - m_use);
#pragma unroll
                for (int r = 0; r < 16; r += 4) { ps0 += p[r]; ps1 += p[r + 1]; ps2 += p[r + 2]; ps3 += p[r + 3]; }
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    u32x4 w;
                    w[0] = ppa_cvt_pk(p[tt * 8 + 0], p[tt * 8 + 1]);
                    w[1] = ppa_cvt_pk(p[tt * 8 + 2], p[tt * 8 + 3]);
                    w[2] = ppa_cvt_pk(p[tt * 8 + 4], p[tt * 8 + 5]);
                    w[3] = ppa_cvt_pk(p[tt * 8 + 6], p[tt * 8 + 7]);
                    pf[blk][tt] = __builtin_bit_cast(bf16x8, w);
                }
            }
            l_run += (ps0 + ps1) + (ps2 + ps3);
            read_v(t);
            if (more) read_k(t + 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        // ================= M_t : matrix pipe only
        if (active) {
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int blk = 0; blk < 2; ++blk)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int d = 0; d < NDB; ++d)
                        o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf[d][blk * 2 + tt], pf[blk][tt], o[d], 0, 0, 0);
            if (more) qk();
            __builtin_amdgcn_s_setprio(0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();                      // pairs with group 1's last barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // trailing prefetches must land before the LDS is released

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

// returns false when this kernel does not apply (head_dim != 64 or a kv slab beyond 2 GiB): the caller goes on to
// attention_bf16.hip
bool gar_attn_pp_bf16_try(const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv, int hd, int q_len,
                          int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev, hipStream_t s) {
    if (hd != 64 || (int64_t)kv_stride * hd * 2 >= (int64_t)1 << 31) return false;
    if (q_len < 256) return false;                      // short query blocks (decode-like calls) keep the 128-row kernel
    dim3 grid(((q_len + 255) / 256) * Hq * B), block(512);
    constexpr int lds = 4 * (64 * 128 + 64 * 128);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pp_bf16_kernel<true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_pp_bf16_kernel<false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        attr_set = true;
    }
    if (causal)
        hipLaunchKernelGGL((attn_pp_bf16_kernel<true>), grid, block, lds, s, (const bf16_t*)Q, (const bf16_t*)K,
                           (const bf16_t*)Vt, (bf16_t*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev);
    else
        hipLaunchKernelGGL((attn_pp_bf16_kernel<false>), grid, block, lds, s, (const bf16_t*)Q, (const bf16_t*)K,
                           (const bf16_t*)Vt, (bf16_t*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev);
    return true;
}
