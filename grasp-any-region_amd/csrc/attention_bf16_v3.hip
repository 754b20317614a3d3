// bf16 flash attention, v3 (prefill / ViT): the lane-local S^T -> softmax -> P -> O^T scheme of attention_bf16.hip,
// software-pipelined inside every wave so that the matrix pipe and the VALU work on different kv tiles at the same time:
//
//   iteration t of a wave (one 64-kv tile, 32 query rows):
//     [rare branch]  rescale O / l if tile t's row max (found in the previous iteration) outgrew the running max
//     region A       8 x QK^T MFMAs of tile t+1  (K fragments of tile t+1 from LDS)
//                    || exp2 / row sum / bf16 pack of tile t's scores -> P_t            (~112 VALU, 32 of them v_exp)
//     region B       8 x PV MFMAs of tile t      (Vt fragments of tile t from LDS, P_t from registers)
//                    || causal / tail masking and row max of tile t+1's scores          (~35 VALU)
//   so a wave always has MFMAs in flight while it runs its softmax, instead of QK^T -> softmax -> PV in series (v2:
//   the matrix pipes were 42 % busy, rocprofv3 SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE, profiles/r2_pmc_attention.json).
//   The two score registers sets alternate roles (the loop body is instantiated twice: no register moves).
//
//   K / Vt tiles: `buffer_load_dwordx4 ... lds` (lane-linear LDS image, XOR swizzle on the SOURCE offset) into rings of
//   NSLOT tiles each; tile t+1's K and tile t's Vt were issued NSLOT-1 iterations before they are read and are waited for
//   with a COUNTED vmcnt (in-order retirement: "all but the newest (NSLOT-2) groups"), then one raw s_barrier per iteration
//   publishes them to the four waves and doubles as the WAR fence for the slot the next DMA group overwrites.
//   head_dim 64: 3 slots (48 KiB, two workgroups per CU); head_dim 128: 2 slots (64 KiB).
#include <stdlib.h>

#include "common.h"

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define V3_RESCALE_THR 6.0f   // log2 domain: P <= 2^6 between rescales

typedef __bf16 v3_bf16v2_t __attribute__((ext_vector_type(2)));
typedef float v3_f32v2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned int v3_cvt_pk(float lo, float hi) {
    v3_f32v2_t v = {lo, hi};
    return __builtin_bit_cast(unsigned int, __builtin_convertvector(v, v3_bf16v2_t));
}
template <int RS> __device__ __forceinline__ int v3_key(int row) { return RS == 128 ? ((row >> 1) & 7) : (row & 15); }

template <int HD, bool CAUSAL>
__global__ __launch_bounds__(256, HD == 64 ? 2 : 1) void attn_bf16_v3_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ Vt, bf16_t* __restrict__ O,
                                                              int Hq, int Hkv, int q_len, int q_pad, int kv_len_arg,
                                                              int kv_stride, const int32_t* __restrict__ kv_len_dev) {
    constexpr int KRS = HD * 2;                 // K tile row bytes
    constexpr int KT = 64 * KRS;                // K tile bytes   [64 kv][HD]
    constexpr int VT = HD * 128;                // Vt tile bytes  [HD][64 kv]
    constexpr int NKD = HD / 16;                // QK^T k-steps
    constexpr int NDB = HD / 32;                // O^T row blocks
    constexpr int KI = KT / 4096;               // K DMA instructions per wave per tile (1 KiB each)
    constexpr int VI = VT / 4096;
    constexpr int NSLOT = HD == 64 ? 3 : 2;
    constexpr int GROUP = KI + VI;              // DMA instructions per wave per iteration
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [NSLOT] K tiles | [NSLOT] Vt tiles
    char* const kring = smem;
    char* const vring = smem + NSLOT * KT;
    const int kv_len = kv_len_dev ? kv_len_dev[0] : kv_len_arg;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    // XCD-aware 1-D grid (see attention_bf16.hip): every XCD gets one contiguous chunk of the (batch, head, q-block) list
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
    const bf16_t* Vp = Vt + (((int64_t)b * Hkv + kvh) * HD) * (int64_t)kv_stride;
    const unsigned slab = (unsigned)kv_stride * HD * 2u;
    const __amdgpu_buffer_rsrc_t rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)slab, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)slab, 0x00020000);

    int voffK, voffV;
    if (HD == 64) {          // piece = 8 rows x 128 B
        const int row = wave * 8 + (lane >> 3);
        voffK = row * 128 + (((lane & 7) ^ v3_key<128>(row)) << 4);
    } else {                 // piece = 4 rows x 256 B
        const int row = wave * 4 + (lane >> 4);
        voffK = row * 256 + (((lane & 15) ^ v3_key<256>(row)) << 4);
    }
    {
        const int row = wave * 8 + (lane >> 3);      // Vt piece = 8 d-rows x 128 B (64 kv)
        voffV = (int)((unsigned)row * (unsigned)kv_stride * 2u) + (((lane & 7) ^ v3_key<128>(row)) << 4);
    }
    auto dma_k = [&](int t, int slot) {
        char* ks = kring + slot * KT;
        const unsigned kbase = (unsigned)t * 64u * KRS;
#pragma unroll
        for (int i = 0; i < KI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsK, LDS_AS(ks + (i * 4 + wave) * 1024), 16,
                                                     voffK + (int)(kbase + (unsigned)i * 4096u), 0, 0, 0);
    };
    auto dma_v = [&](int t, int slot) {
        char* vs = vring + slot * VT;
        const unsigned vbase = (unsigned)t * 128u;
#pragma unroll
        for (int i = 0; i < VI; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsV, LDS_AS(vs + (i * 4 + wave) * 1024), 16,
                                                     voffV + (int)(vbase + (unsigned)i * 32u * (unsigned)kv_stride * 2u),
                                                     0, 0, 0);
    };

    int kv_end = kv_len;
    if (CAUSAL) kv_end = min(kv_len, qb * 128 + 127 + coff + 1);
    const int ntiles = (kv_end + 63) / 64;                       // tiles of this workgroup (>= 1: kv_len >= 1)
    // tiles this WAVE computes (causal: the waves of a workgroup stop at different tiles; they keep DMAs and barriers)
    const bool wave_active = q0 < q_len;
    int wtiles = wave_active ? ntiles : 0;
    if (CAUSAL && wave_active) wtiles = min(ntiles, (min(kv_len, q0 + 31 + coff + 1) + 63) / 64);

    // DMA groups: the prologue brings K(0), then group j = {K(j+1), Vt(j)}; NSLOT-1 groups are in flight at a time.
    // Tiles past the end are issued all the same (nobody reads them): the counted waits stay uniform.
    // Q fragments (B operand): Q[q0 + l31][16 kd + 8h .. +8] — issued BEFORE the DMAs: vmcnt retires in order, so the
    // wait for these registers does not drain the tile prefetches behind them
    bf16x8 qf[NKD];
    {
        const int qrow = min(q0 + l31, q_pad - 1);
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd)
            qf[kd] = *reinterpret_cast<const bf16x8*>(Qp + (int64_t)qrow * HD + kd * 16 + h * 8);
    }
    dma_k(0, 0);
#pragma unroll
    for (int j = 0; j < NSLOT - 1; ++j) {
        dma_k(j + 1, (j + 1) % NSLOT);
        dma_v(j, j % NSLOT);
    }
    f32x16 o[NDB];
#pragma unroll
    for (int d = 0; d < NDB; ++d)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[d][r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);   // bits 2<->3 swapped (see attention.hip)
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int koff[2], kkey[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const int row = blk * 32 + prow;
        koff[blk] = row * KRS;
        kkey[blk] = v3_key<KRS>(row);
    }

    // S^T of tile t: s[blk] register r <-> kv = 64 t + 32 blk + 16 (r>>3) + 8 h + (r&7), query q0 + l31
    auto qk = [&](int slot, f32x16 (&s)[2]) {
        const char* ks = kring + slot * KT;
#pragma unroll
        for (int kd = 0; kd < NKD; ++kd) {
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const bf16x8 kf = *reinterpret_cast<const bf16x8*>(ks + koff[blk] + (((kd * 2 + h) ^ kkey[blk]) << 4));
                s[blk] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[kd], kd == 0 ? zero16 : s[blk], 0, 0, 0);
            }
        }
    };
    // causal / tail masking of a score tile (only tiles that cross the diagonal or the end of the kv range)
    auto needs_mask = [&](int t) { return (t * 64 + 64 > kv_len) || (CAUSAL && t * 64 + 63 > q0 + coff); };   // wave-uniform
    auto apply_mask = [&](int t, f32x16 (&s)[2]) {
        const int kv0 = t * 64;
        const int qi = q0 + l31;
        const int lim = CAUSAL ? min(kv_len - 1, qi + coff) : kv_len - 1;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int kv = kv0 + blk * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                s[blk][r] = kv <= lim ? s[blk][r] : -INFINITY;
            }
    };
    // row max over both half-waves
    auto row_max = [&](f32x16 (&s)[2]) -> float {
        float mx = -INFINITY;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[blk][r]);
        return fmaxf(mx, __shfl_xor(mx, 32, 64));
    };
    // deferred rescale: only when some row's running max would grow by more than 2^THR (NaN = -inf - -inf lands here too)
    auto maybe_rescale = [&](float mx) {
        if (!__all(mx - m_run <= V3_RESCALE_THR)) {
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
    };

    // ---- prologue: S_0 and its row max
    f32x16 sa[2], sb[2];
    float mx_cur = -INFINITY;
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 1) * GROUP) : "memory");          // K(0) landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();
    if (wtiles > 0) {
        qk(0, sa);
        if (needs_mask(0)) apply_mask(0, sa);
        mx_cur = row_max(sa);
    }

    // exp2 / row sum / bf16 pack of tile t's scores -> P_t (B operand fragments of the PV MFMAs)
    auto softmax_p = [&](f32x16 (&sc)[2], float m_use, bf16x8 (&pf)[2][2]) {
        float ps = 0.f;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { p[r] = __builtin_amdgcn_exp2f(sc[blk][r] - m_use); ps += p[r]; }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                u32x4 w;
                w[0] = v3_cvt_pk(p[tt * 8 + 0], p[tt * 8 + 1]);
                w[1] = v3_cvt_pk(p[tt * 8 + 2], p[tt * 8 + 3]);
                w[2] = v3_cvt_pk(p[tt * 8 + 4], p[tt * 8 + 5]);
                w[3] = v3_cvt_pk(p[tt * 8 + 6], p[tt * 8 + 7]);
                pf[blk][tt] = __builtin_bit_cast(bf16x8, w);
            }
        }
        l_run += ps;
    };
    auto pv = [&](int slot, bf16x8 (&pf)[2][2]) {
        const char* vs = vring + slot * VT;
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int d = 0; d < NDB; ++d) {
                    const int row = d * 32 + l31;
                    const int c = (blk * 2 + tt) * 2 + h;
                    const bf16x8 vf = *reinterpret_cast<const bf16x8*>(vs + row * 128 + ((c ^ v3_key<128>(row)) << 4));
                    o[d] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf[blk][tt], o[d], 0, 0, 0);
                }
    };
    constexpr int NQK = 2 * NKD, NPV = 4 * NDB;          // MFMAs of region A / region B
    // one iteration: `sc` holds tile t's scores (row max mx_cur), `sn` receives tile t+1's
    auto body = [&](int t, f32x16 (&sc)[2], f32x16 (&sn)[2]) {
        // group t = {K(t+1), Vt(t)} must have landed: all but the newest NSLOT-2 groups of this wave, then the barrier
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NSLOT - 2) * GROUP) : "memory");
        __builtin_amdgcn_s_barrier();
        // slots of K(t) and Vt(t-1) were last read in iteration t-1: every wave is past that now
        dma_k(t + NSLOT, t % NSLOT);
        dma_v(t + NSLOT - 1, (t + NSLOT - 1) % NSLOT);
        if (t >= wtiles) return;                                                     // wave-uniform (causal tail / idle wave)
        maybe_rescale(mx_cur);
        const float m_use = m_run == -INFINITY ? 0.f : m_run;
        bf16x8 pf[2][2];
        if (t + 1 < wtiles) {
            // ---- region A: QK^T of tile t+1 || exp / sum / pack of tile t (one basic block; MFMAs spread over the VALU)
            qk((t + 1) % NSLOT, sn);
            softmax_p(sc, m_use, pf);
            // schedule: all K fragment reads up front, a first slice of VALU to cover their latency, then one MFMA per
            // slice of the remaining softmax VALU
            __builtin_amdgcn_sched_group_barrier(0x100, NQK, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
#pragma unroll
            for (int i = 0; i < NQK; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (100 + NQK - 1) / NQK, 0);
            }
            if (needs_mask(t + 1)) apply_mask(t + 1, sn);
            // ---- region B: PV of tile t || row max of tile t+1
            pv(t % NSLOT, pf);
            mx_cur = row_max(sn);
            __builtin_amdgcn_sched_group_barrier(0x100, NPV, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
#pragma unroll
            for (int i = 0; i < NPV; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, (32 + NPV - 1) / NPV, 0);
            }
        } else {                                                                     // the wave's last tile
            softmax_p(sc, m_use, pf);
            pv(t % NSLOT, pf);
        }
    };
    for (int t = 0; t < ntiles; t += 2) {
        body(t, sa, sb);
        if (t + 1 < ntiles) body(t + 1, sb, sa);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // the unused trailing prefetches must land before the LDS is released

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

// returns false when this kernel does not apply (the caller goes on to attention_bf16.hip's v2 / attention.hip)
bool gar_attn_bf16_v3_try(const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv, int hd, int q_len,
                          int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev, hipStream_t s) {
    static const int mode = [] { const char* e = getenv("GAR_ATTN_V3"); return e ? atoi(e) : 1; }();
    if (!mode) return false;
    if ((int64_t)kv_stride * hd * 2 >= (int64_t)1 << 31) return false;
    if (hd != 64 && hd != 128) return false;
    dim3 grid(((q_len + 127) / 128) * Hq * B), block(256);
    const int nslot = hd == 64 ? 3 : 2;
    const int lds = nslot * (64 * hd * 2 + hd * 128);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bf16_v3_kernel<128, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 * 128 * 2 + 128 * 128));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bf16_v3_kernel<128, false>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (64 * 128 * 2 + 128 * 128));
        attr_set = true;
    }
#define LAUNCH_V3(HD_, C_)                                                                                            \
    hipLaunchKernelGGL((attn_bf16_v3_kernel<HD_, C_>), grid, block, lds, s, (const bf16_t*)Q, (const bf16_t*)K,       \
                       (const bf16_t*)Vt, (bf16_t*)O, Hq, Hkv, q_len, q_pad, kv_len, kv_stride, kv_len_dev)
    if (hd == 64) { if (causal) LAUNCH_V3(64, true); else LAUNCH_V3(64, false); }
    else { if (causal) LAUNCH_V3(128, true); else LAUNCH_V3(128, false); }
#undef LAUNCH_V3
    return true;
}
