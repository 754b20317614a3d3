// bf16 flash attention, v4 (ViT tiles / Llama causal prefill; head_dim 64, row-major V): the frame of the CDNA guide's
// 4-wave persistent attention instead of v2's 3-waves-per-SIMD frame (attention_bf16.hip) —
//   * a workgroup = 4 waves = ONE WAVE PER SIMD with the whole 512-register file; a wave owns 64 query rows (two 32-row
//     q-blocks), a workgroup a 256-row Q block: every staged K / V tile and every LDS fragment read serves twice the query
//     rows of v2 (half the tile traffic and half the ds_reads per MFMA);
//   * the work list (batch, head, Q block) is walked by PERSISTENT workgroups (one per CU, XCD-aware order: the Q blocks / query
//     heads that share a kv slab run on the same XCD at the same time); K / V tiles arrive by LDS-DMA into a ring of V4_NS
//     stages as ONE continuous stream across work items — the tile fetched in an iteration is V4_AHEAD tiles ahead, whatever
//     item it belongs to — and the NEXT item's Q rows (and folded prefix key / value row) are DMA'd into LDS while the current
//     item computes: no per-item prologue on the critical path (v2: a third of its time);
//   * with one wave per SIMD nothing hides a wave's softmax but its OWN matrix instructions (tools/interleave_probe.hip: a wave's
//     VALU issues in the shadow of its own MFMAs, up to ~5 per 32x32x16), so the kv loop is software-pipelined in the wave:
//     iteration j = [ PV(j-1) | softmax(j, q-block 0) ] [ QK^T(j+1) | softmax(j, q-block 1) ], 16 MFMAs and 80 VALU per phase in
//     fenced bundles; LDS fragment reads are inline asm with counted lgkmcnt (the compiler would put `s_waitcnt vmcnt(0)` in
//     front of a transposing read that follows an LDS-DMA, which would drain the prefetch ring every tile);
//   * same arithmetic as v2: S^T = K Q^T with the -m accumulator start, lazy running max (exact pass on the first tile, on
//     masked tiles and when a row sum reaches 2^H16_MAX_LOG2), P packed in the lane, O^T += V^T P through ds_read_b64_tr_b16;
//     O leaves through a per-wave LDS transposition as whole 128-byte rows.
// Semantics: softmax(q k^T) v of timm Eva's SDPA (modeling_perception_lm.py:210-214, non-causal, one tile per batch item) and of
// flash-attn-2's causal GQA prefill (modeling_gar.py:40-43), q pre-scaled by scale * log2(e) by the qkv epilogues.
#include <stdlib.h>

#include <type_traits>

#include "common.h"

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
typedef short tr4_t __attribute__((ext_vector_type(4)));

#define V4_NS 5                           // ring stages: tile g of a workgroup's stream lives in stage g % V4_NS
#define V4_AHEAD 3                        // an iteration whose softmax tile is g fetches tile g + V4_AHEAD
#define V4_STAGE 16384                    // [K tile 64 x 128 B | V tile 64 x 128 B]
#define V4_QW 10240                       // per wave: 64 Q rows x 128 B + the folded prefix key row + value row (1 KiB pieces)
#define V4_QBUF (V4_NS * V4_STAGE)
#define V4_OBUF (V4_QBUF + 4 * V4_QW)
#define V4_LDS (V4_OBUF + 4 * 8192)       // 155648 B of the CU's 160 KiB
#define V4_RESCALE_THR 6.0f               // log2 domain

struct v4_args {
    const bf16_t* Q;            // [B, Hq, q_pad, 64]
    const bf16_t* K;            // [B, Hkv, kv_stride, 64]
    const bf16_t* V;            // [B, Hkv, kv_stride, 64] row-major
    bf16_t* O;                  // [B * q_total, Hq * 64]
    const int32_t* kv_len_dev;
    const int32_t* kv_start;
    int B, Hq, Hkv, q_row0, q_len, q_total, q_pad, kv_len, kv_stride, kv_prefix, nqb;
};

template <int F, int N, class Fn>
__device__ __forceinline__ void v4_for(Fn&& fn) {
    if constexpr (F < N) {
        fn(std::integral_constant<int, F>{});
        v4_for<F + 1, N>(fn);
    }
}

// LDS fragment reads the compiler does not see (no vmcnt(0) in front of them, no lgkmcnt bookkeeping): counted waits below.
template <int OFF>
__device__ __forceinline__ bf16x8 v4_lds_b128(unsigned addr) {
    bf16x8 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
    return d;
}
template <int OFF>
__device__ __forceinline__ tr4_t v4_lds_tr(unsigned addr) {
    tr4_t d;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
    return d;
}
// wait until at most N of this wave's LDS operations are outstanding; the named registers are not consumed before it
template <int N>
__device__ __forceinline__ void v4_wait_lgkm(bf16x8& x) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(x) : "i"(N));
}
template <int N>
__device__ __forceinline__ void v4_wait_lgkm2(tr4_t& x, tr4_t& y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(x), "+v"(y) : "i"(N));
}
__device__ __forceinline__ void v4_wait_vm(int n) {          // n wave-uniform: at most n VMEM operations outstanding (rounded down)
    if (n >= 28) asm volatile("s_waitcnt vmcnt(28)" ::: "memory");
    else if (n >= 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
    else if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else if (n >= 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if (n >= 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if (n >= 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (n >= 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
#define V4_FENCE __builtin_amdgcn_sched_barrier(0)

template <bool CAUSAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bf16_v4_kernel(const v4_args a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int PFX = CAUSAL ? 0 : a.kv_prefix;
    const int kv_len = (a.kv_len_dev ? a.kv_len_dev[0] : a.kv_len) - PFX;
    const int coff = kv_len - a.q_total;                 // causal: kv <= q + coff
    const int q_end = a.q_row0 + a.q_len;
    const int G = gridDim.x;
    const int n_items = a.B * a.Hq * a.nqb;
    const int gsz = a.Hq / a.Hkv;
    const unsigned slab = (unsigned)a.kv_stride * 128u;

    // ---- work list. Item index `it`: non-causal = ((b, head), Q block) — the Q blocks of a (b, head) share its K / V slab;
    // causal = (Q block from the LAST one down: heavy items first, (b, head)) — the heads of a GQA group are neighbours.
    // Workgroup L runs on XCD L % 8 (round-robin dispatch): in step k it takes item k G + (L % 8) (G / 8) + L / 8, so every XCD
    // works on G / 8 CONSECUTIVE items at a time and re-reads shared slabs from its own L2.
    struct Item {
        int valid, b, head, qb, t_lo, ntiles, kv_lo;
        __amdgpu_buffer_rsrc_t rsK, rsV, rsQ;
    };
    auto decode = [&](int k, Item& I) __attribute__((always_inline)) {
        const int L = blockIdx.x;
        const int it = (G & 7) == 0 ? k * G + (L & 7) * (G >> 3) + (L >> 3) : k * G + L;
        I.valid = it < n_items;
        const int itc = I.valid ? it : 0;
        int b, head, qb;
        if (CAUSAL) {
            const int per = a.B * a.Hq;
            qb = a.nqb - 1 - itc / per;
            const int rem = itc % per;
            b = rem / a.Hq;
            head = rem % a.Hq;
        } else {
            qb = itc % a.nqb;
            const int bh = itc / a.nqb;
            b = bh / a.Hq;
            head = bh % a.Hq;
        }
        I.b = b; I.head = head; I.qb = qb;
        const int kvh = head / gsz;
        const int kv_lo = a.kv_start ? max(min(a.kv_start[b], kv_len - 1), 0) : 0;
        I.kv_lo = kv_lo;
        I.t_lo = kv_lo >> 6;
        int kv_end = kv_len;
        if (CAUSAL) kv_end = min(kv_len, max(min(a.q_row0 + qb * 256 + 255, q_end - 1) + coff, kv_lo) + 1);
        I.ntiles = (kv_end + 63) >> 6;
        const bf16_t* Kp = a.K + ((int64_t)b * a.Hkv + kvh) * (int64_t)a.kv_stride * 64;
        const bf16_t* Vp = a.V + ((int64_t)b * a.Hkv + kvh) * (int64_t)a.kv_stride * 64;
        const bf16_t* Qp = a.Q + ((int64_t)b * a.Hq + head) * (int64_t)a.q_pad * 64;
        I.rsK = __builtin_amdgcn_make_buffer_rsrc((void*)Kp, 0, (int)slab, 0x00020000);
        I.rsV = __builtin_amdgcn_make_buffer_rsrc((void*)Vp, 0, (int)slab, 0x00020000);
        I.rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qp, 0, a.q_pad * 128, 0x00020000);
    };

    // ---- LDS-DMA: a 64-row tile = 8 pieces of 8 rows x 128 B (1 KiB, lane-linear in LDS); wave w brings pieces w and 4 + w of K
    // and of V. The XOR swizzle of the 16-byte chunks is applied to the SOURCE offset (K: key (row >> 1) & 7 -> conflict-free
    // b128 fragment reads; V: key 4 ((row >> 1) & 1) -> the four rows of a transposing-read block cover all banks).
    const int drow = wave * 8 + (lane >> 3);
    const int voffK = drow * 128 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);
    const int voffV = drow * 128 + (((lane & 7) ^ (((drow >> 1) & 1) << 2)) << 4);
    auto issue_tile = [&](const Item& I, int t, int stage) __attribute__((always_inline)) {
        char* ks = smem + stage * V4_STAGE;
        const int base = (t * 64 + PFX) * 128;
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(I.rsK, LDS_AS(ks + (i * 4 + wave) * 1024), 16, voffK + base + i * 4096, 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(I.rsV, LDS_AS(ks + 8192 + (i * 4 + wave) * 1024), 16, voffV + base + i * 4096, 0, 0,
                                                     0);
    };
    // the wave's own 64 Q rows of item I (row-major image, no swizzle: read once per item) + the prefix key / value row
    char* const qbuf = smem + V4_QBUF + wave * V4_QW;
    char* const obuf = smem + V4_OBUF + wave * 8192;
    const int voffQ = (lane >> 3) * 128 + ((lane & 7) << 4);
    const int QOPS = 8 + (PFX ? 2 : 0);
    auto issue_q = [&](const Item& I) __attribute__((always_inline)) {
        const int row0 = a.q_row0 + I.qb * 256 + wave * 64;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(I.rsQ, LDS_AS(qbuf + p * 1024), 16, voffQ + (row0 + p * 8) * 128, 0, 0, 0);
        if (PFX) {          // row 0 of the slab, 8 chunks (replicated over the 8 row slots of the piece)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(I.rsK, LDS_AS(qbuf + 8192), 16, (lane & 7) << 4, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(I.rsV, LDS_AS(qbuf + 9216), 16, (lane & 7) << 4, 0, 0, 0);
        }
    };

    // ---- per-lane fragment addresses (bytes inside a stage): K rows in the permuted order that makes accumulator registers
    // 8t .. 8t+7 the kv slice the PV MFMA wants (attention.hip); V through the transposing read (attention_bf16.hip)
    const int prow = (l31 & 0x13) | ((l31 & 4) << 1) | ((l31 & 8) >> 1);
    const int kkey = (prow >> 1) & 7;
    unsigned kaddr0[4];
#pragma unroll
    for (int kd = 0; kd < 4; ++kd) kaddr0[kd] = lds0 + prow * 128 + (((kd * 2 + h) ^ kkey) << 4);
    unsigned vaddr0[2];
    {
        const int i = lane & 15, r = 8 * h + (i >> 2);
        const int col = 16 * ((lane >> 4) & 1) + 4 * (i & 3);
        const int key4 = ((r >> 1) & 1) << 2;
        const int vtr = r * 128 + ((((col >> 3) ^ key4) << 4) | ((col & 7) << 1));
        vaddr0[0] = lds0 + 8192 + vtr;
        vaddr0[1] = lds0 + 8192 + (vtr ^ 64);
    }

    // ---- wave state of the current item
    f32x16 o[2][2];                  // [q-block][d-block]
    f32x16 negm[2];                  // -m of the q-block's row (16 equal registers: the C operand of the first QK^T MFMA)
    f32x16 s[2][2][2];               // [parity][q-block][kv block]: scores of the tile being exponentiated / of the next one
    u32x4 pf[2][2][2][2];            // [parity][q-block][kv block][16-kv step]: P as PV B operands
    bf16x8 qf[2][4];
    float m_run[2], l_run[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // QK^T of the tile in `stage` into s[P]: un-overlapped form (first tile of an item, around masked tiles)
    auto qk_plain = [&](auto pc, int stage) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
        unsigned ka[4];
#pragma unroll
        for (int kd = 0; kd < 4; ++kd) ka[kd] = kaddr0[kd] + stage * V4_STAGE;
        bf16x8 kf[2][4];
        v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value, kd = f >> 1, kb = f & 1;
            kf[kb][kd] = v4_lds_b128<kb * 4096>(ka[kd]);
        });
        v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value, kd = f >> 1, kb = f & 1;
            v4_wait_lgkm<7 - f>(kf[kb][kd]);
            s[P][0][kb] = MFMA_32x32x16(kf[kb][kd], qf[0][kd], kd == 0 ? negm[0] : s[P][0][kb]);
            s[P][1][kb] = MFMA_32x32x16(kf[kb][kd], qf[1][kd], kd == 0 ? negm[1] : s[P][1][kb]);
        });
    };
    // O^T += V^T P of the tile in `stage` with pf[P]
    auto pv_plain = [&](auto pc, int stage) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value;
        unsigned va[2] = {vaddr0[0] + (unsigned)(stage * V4_STAGE), vaddr0[1] + (unsigned)(stage * V4_STAGE)};
        tr4_t lo[8], hi[8];
        v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1, db = f & 1;
            lo[f] = v4_lds_tr<(kb * 32 + tt * 16) * 128>(va[db]);
            hi[f] = v4_lds_tr<(kb * 32 + tt * 16 + 4) * 128>(va[db]);
        });
        v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1, db = f & 1;
            v4_wait_lgkm2<14 - 2 * f>(lo[f], hi[f]);
            const bf16x8 vf = {lo[f][0], lo[f][1], lo[f][2], lo[f][3], hi[f][0], hi[f][1], hi[f][2], hi[f][3]};
            o[0][db] = MFMA_32x32x16(vf, __builtin_bit_cast(bf16x8, pf[P][0][kb][tt]), o[0][db]);
            o[1][db] = MFMA_32x32x16(vf, __builtin_bit_cast(bf16x8, pf[P][1][kb][tt]), o[1][db]);
        });
    };
    // p = exp2(s) (s carries -m), row sum, bf16 pack of q-block qb of s[P] -> pf[P][qb]; returns this lane's partial row sum
    auto exp_pack = [&](auto pc, auto qc) __attribute__((always_inline)) -> float {
        constexpr int P = decltype(pc)::value, qb = decltype(qc)::value;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[P][qb][kb][r]);
#pragma unroll
            for (int r = 0; r < 16; r += 2) { ps0 += p[r]; ps1 += p[r + 1]; }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                pf[P][qb][kb][tt] = u32x4{pack_bf2(p[tt * 8 + 0], p[tt * 8 + 1]), pack_bf2(p[tt * 8 + 2], p[tt * 8 + 3]),
                                          pack_bf2(p[tt * 8 + 4], p[tt * 8 + 5]), pack_bf2(p[tt * 8 + 6], p[tt * 8 + 7])};
        }
        return ps0 + ps1;
    };
    // exact softmax step of q-block qb on s[P] (first tile, masked tiles, lazy-max overflow): optional mask, exact tile max,
    // re-base of the running max (O, l, the scores — and the NEXT tile's scores in s[P ^ 1] when they were computed against
    // the old max), exp / pack. All of PV up to the previous tile must be in O.
    auto exact_step = [&](auto pc, auto qc, bool need_mask, int kv0, int q0w, int kv_lo, bool shift_next) __attribute__((always_inline)) {
        constexpr int P = decltype(pc)::value, qb = decltype(qc)::value;
        if (need_mask) {
            const int qi = q0w + qb * 32 + l31;
            const int lim = CAUSAL ? min(kv_len - 1, max(qi + coff, kv_lo)) : kv_len - 1;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + kb * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                    s[P][qb][kb][r] = (kv <= lim && kv >= kv_lo) ? s[P][qb][kb][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[P][qb][kb][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));          // relative to m_base = -negm (0 while m_run is still -inf)
        // (m_base - m_run) is 0 once the row has a finite max and +inf before: NaN / +inf also land in the branch
        if (!__all(mx + (-negm[qb][0] - m_run[qb]) <= V4_RESCALE_THR)) {
            const float m_base = -negm[qb][0];
            const float m_new = fmaxf(m_run[qb], mx + m_base);
            const float m_nu = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run[qb] - m_nu);
            const float shift = m_base - m_nu;
            m_run[qb] = m_new;
            l_run[qb] *= alpha;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][d][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[P][qb][kb][r] += shift;
            if (shift_next) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                    for (int r = 0; r < 16; ++r) s[P ^ 1][qb][kb][r] += shift;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[qb][r] = -m_nu;
        }
        l_run[qb] += exp_pack(pc, qc);
    };

    // ---- the fused iteration of the kv loop (unmasked tile j with PV(j-1) pending and tile j+1 to score):
    //   phase A: 16 MFMAs of O^T += V^T(j-1) P(j-1)  |  exp / sum / pack of q-block 0 of tile j
    //   phase B: 16 MFMAs of S^T(j+1) = K(j+1) Q^T    |  exp / sum / pack of q-block 1 of tile j
    // in eight fenced bundles per phase: [fragment read 2 bundles ahead | counted wait | 2 MFMAs | 4 exp, 4 add, 2 cvt_pk]; the
    // four DMA instructions of the tile V4_AHEAD ahead ride in the first four bundles.
    auto fused = [&](auto pc, int st_prev, int st_next, bool dma, const Item& DI, int dt, int dstage) __attribute__((always_inline)) -> bool {
        constexpr int P = decltype(pc)::value;
        float ps[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
        char* dks = smem + dstage * V4_STAGE;
        const int dbase = (dt * 64 + PFX) * 128;
        // softmax slice m (0..7) of q-block qb: elements 4m .. 4m+3 of the 32 scores of this lane
        auto sm_slice = [&](auto qc, auto mc) __attribute__((always_inline)) {
            constexpr int qb = decltype(qc)::value, m = decltype(mc)::value, kb = m >> 2, r0 = (m & 3) * 4;
            const float p0 = __builtin_amdgcn_exp2f(s[P][qb][kb][r0]), p1 = __builtin_amdgcn_exp2f(s[P][qb][kb][r0 + 1]);
            const float p2 = __builtin_amdgcn_exp2f(s[P][qb][kb][r0 + 2]), p3 = __builtin_amdgcn_exp2f(s[P][qb][kb][r0 + 3]);
            ps[qb][0] += p0;
            ps[qb][1] += p1;
            ps[qb][0] += p2;
            ps[qb][1] += p3;
            constexpr int tt = r0 >> 3, w0 = (r0 & 7) >> 1;
            pf[P][qb][kb][tt][w0] = pack_bf2(p0, p1);
            pf[P][qb][kb][tt][w0 + 1] = pack_bf2(p2, p3);
        };
        // ---- phase A
        {
            unsigned va[2] = {vaddr0[0] + (unsigned)(st_prev * V4_STAGE), vaddr0[1] + (unsigned)(st_prev * V4_STAGE)};
            tr4_t lo[8], hi[8];
            v4_for<0, 2>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1, db = f & 1;
                lo[f] = v4_lds_tr<(kb * 32 + tt * 16) * 128>(va[db]);
                hi[f] = v4_lds_tr<(kb * 32 + tt * 16 + 4) * 128>(va[db]);
            });
            v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1, db = f & 1;
                V4_FENCE;
                if constexpr (f + 2 < 8) {
                    constexpr int f2 = f + 2, st2 = f2 >> 1, kb2 = st2 >> 1, tt2 = st2 & 1, db2 = f2 & 1;
                    lo[f2] = v4_lds_tr<(kb2 * 32 + tt2 * 16) * 128>(va[db2]);
                    hi[f2] = v4_lds_tr<(kb2 * 32 + tt2 * 16 + 4) * 128>(va[db2]);
                }
                if (f < 4 && dma) {         // one DMA instruction of the tile V4_AHEAD ahead per bundle
                    if (f < 2)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(DI.rsK, LDS_AS(dks + ((f & 1) * 4 + wave) * 1024), 16,
                                                                 voffK + dbase + (f & 1) * 4096, 0, 0, 0);
                    else
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(DI.rsV, LDS_AS(dks + 8192 + ((f & 1) * 4 + wave) * 1024), 16,
                                                                 voffV + dbase + (f & 1) * 4096, 0, 0, 0);
                }
                v4_wait_lgkm2<(f + 2 < 8) ? 4 : (f == 6 ? 2 : 0)>(lo[f], hi[f]);
                const bf16x8 vf = {lo[f][0], lo[f][1], lo[f][2], lo[f][3], hi[f][0], hi[f][1], hi[f][2], hi[f][3]};
                o[0][db] = MFMA_32x32x16(vf, __builtin_bit_cast(bf16x8, pf[P ^ 1][0][kb][tt]), o[0][db]);
                o[1][db] = MFMA_32x32x16(vf, __builtin_bit_cast(bf16x8, pf[P ^ 1][1][kb][tt]), o[1][db]);
                sm_slice(std::integral_constant<int, 0>{}, fc);
            });
        }
        // ---- phase B
        {
            unsigned ka[4];
#pragma unroll
            for (int kd = 0; kd < 4; ++kd) ka[kd] = kaddr0[kd] + st_next * V4_STAGE;
            bf16x8 kf[8];
            v4_for<0, 2>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value, kd = f >> 1, kb = f & 1;
                kf[f] = v4_lds_b128<kb * 4096>(ka[kd]);
            });
            v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
                constexpr int f = decltype(fc)::value, kd = f >> 1, kb = f & 1;
                V4_FENCE;
                if constexpr (f + 2 < 8) {
                    constexpr int f2 = f + 2, kd2 = f2 >> 1, kb2 = f2 & 1;
                    kf[f2] = v4_lds_b128<kb2 * 4096>(ka[kd2]);
                }
                v4_wait_lgkm<(f + 2 < 8) ? 2 : (f == 6 ? 1 : 0)>(kf[f]);
                s[P ^ 1][0][kb] = MFMA_32x32x16(kf[f], qf[0][kd], kd == 0 ? negm[0] : s[P ^ 1][0][kb]);
                s[P ^ 1][1][kb] = MFMA_32x32x16(kf[f], qf[1][kd], kd == 0 ? negm[1] : s[P ^ 1][1][kb]);
                sm_slice(std::integral_constant<int, 1>{}, fc);
            });
            V4_FENCE;
        }
        const float t0 = ps[0][0] + ps[0][1], t1 = ps[1][0] + ps[1][1];
        const float lim = (float)(1u << H16_MAX_LOG2);
        const bool ok = __all(t0 < lim && t1 < lim);
        if (ok) {
            l_run[0] += t0;
            l_run[1] += t1;
        }
        return ok;
    };

    // ---- cursors: `cur` = the item being computed (tile t of it), `nxt` = the item after it (its Q is prefetched),
    // `di` = the item the DMA cursor is in (tile dt of it; kd-th item of this workgroup)
    Item cur, nxt, di;
    int kc = 0, kdi = 0;
    decode(0, cur);
    if (!cur.valid) return;
    decode(1, nxt);
    di = cur;
    int dt = cur.t_lo;
    auto advance_dma = [&]() __attribute__((always_inline)) {
        ++dt;
        if (dt >= di.ntiles) {
            ++kdi;
            decode(kdi, di);
            dt = di.t_lo;
        }
    };
    // prologue: the first V4_AHEAD tiles of the stream and the first item's Q rows
    int g = 0;
#pragma unroll
    for (int i = 0; i < V4_AHEAD; ++i) {
        if (di.valid) {
            issue_tile(di, dt, i % V4_NS);
            advance_dma();
        }
    }
    issue_q(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    int t = cur.t_lo;
    bool sv = false, pvp = false;          // s[parity] holds QK^T of tile t / P of tile t - 1 waits for its PV
    int ops_prev_after = 0;                // VMEM operations issued in the previous iteration after its tile DMA
    int ops_since_q = 1 << 20;             // ... issued after the newest Q prefetch
    bool q_inflight = false;
    int q0w = 0;
    bool wave_active = false;
    // item entry: Q fragments out of the wave's LDS rows, the folded prefix key / value as the initial softmax state
    auto enter_item = [&]() __attribute__((always_inline)) {
        q0w = a.q_row0 + cur.qb * 256 + wave * 64;
        wave_active = q0w < q_end;
        const char* qb_ = qbuf;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb)
#pragma unroll
            for (int kd = 0; kd < 4; ++kd)
                qf[qb][kd] = *reinterpret_cast<const bf16x8*>(qb_ + (qb * 32 + l31) * 128 + ((kd * 2 + h) << 4));
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            m_run[qb] = -INFINITY;
            l_run[qb] = 0.f;
            negm[qb] = zero16;
            o[qb][0] = zero16;
            o[qb][1] = zero16;
        }
        if (PFX) {
            // s0 = q . k0 (q carries scale * log2e): this lane holds dims 16 kd + 8 h .. + 8 of its query rows
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float part = 0.f;
#pragma unroll
                for (int kd = 0; kd < 4; ++kd) {
                    const bf16x8 kf = *reinterpret_cast<const bf16x8*>(qb_ + 8192 + ((kd * 2 + h) << 4));
#pragma unroll
                    for (int e = 0; e < 8; ++e) part = __builtin_fmaf(bf2f((bf16_t)qf[qb][kd][e]), bf2f((bf16_t)kf[e]), part);
                }
                m_run[qb] = part + __shfl_xor(part, 32, 64);
                l_run[qb] = h == 0 ? 1.0f : 0.f;             // the two halves' l are added at the end
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qb][r] = -m_run[qb];
                // O0 = v0: register r of d-block d is d index 32 d + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        float v4[4];
                        ld4(reinterpret_cast<const bf16_t*>(qb_ + 9216) + d * 32 + gq * 8 + h * 4, v4);
#pragma unroll
                        for (int r = 0; r < 4; ++r) o[qb][d][gq * 4 + r] = v4[r];
                    }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the Q rows are consumed: the next prefetch may overwrite them
        sv = false;
        pvp = false;
    };
    enter_item();

    // epilogue of the item: O^T fragments -> rows through the wave's LDS piece -> 16-byte stores (8 rows x 128 B per instruction)
    auto store_item = [&]() __attribute__((always_inline)) -> int {
        if (!wave_active) return 0;
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
            const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
            const int row = qb * 32 + l31;
#pragma unroll
            for (int d = 0; d < 2; ++d)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq)
                    *reinterpret_cast<u32x2*>(obuf + row * 128 + (((d * 4 + gq) ^ (row & 7)) << 4) + h * 8) =
                        u32x2{pack_bf2(o[qb][d][gq * 4 + 0] * inv, o[qb][d][gq * 4 + 1] * inv),
                              pack_bf2(o[qb][d][gq * 4 + 2] * inv, o[qb][d][gq * 4 + 3] * inv)};
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int nst = min(8, (q_end - q0w + 7) >> 3);           // wave-uniform
        bf16_t* const Ow = a.O + ((int64_t)cur.b * a.q_total + q0w) * ((int64_t)a.Hq * 64) + cur.head * 64;
        const int rr = lane >> 3, c = lane & 7;
        for (int i = 0; i < nst; ++i) {
            const int row = i * 8 + rr;
            const u32x4 v = *reinterpret_cast<const u32x4*>(obuf + row * 128 + ((c ^ (row & 7)) << 4));
            if (q0w + row < q_end) *reinterpret_cast<u32x4*>(Ow + (int64_t)row * (a.Hq * 64) + c * 8) = v;
        }
        return nst;
    };

    auto iteration = [&](auto pc) __attribute__((always_inline)) -> bool {
        constexpr int P = decltype(pc)::value;
        auto pn = std::integral_constant<int, P ^ 1>{};
        int ops_cur = 0;
        const bool dma = di.valid;
        const Item DI = di;
        const int ddt = dt, dstage = (g + V4_AHEAD) % V4_NS;
        const int st_cur = g % V4_NS, st_prev = (g + V4_NS - 1) % V4_NS, st_next = (g + 1) % V4_NS;
        const int kv0 = t * 64;
        const bool last = t + 1 >= cur.ntiles;
        const bool skip = !wave_active || (CAUSAL && kv0 > max(q0w + 63 + coff, cur.kv_lo));
        const bool next_act = !last && wave_active && !(CAUSAL && kv0 + 64 > max(q0w + 63 + coff, cur.kv_lo));
        const bool need_mask = (kv0 + 64 > kv_len) || kv0 < cur.kv_lo || (CAUSAL && kv0 + 63 > q0w + coff);
        bool dma_done = false;
        if (!skip && sv && pvp && !need_mask && next_act) {
            const bool ok = fused(pc, st_prev, st_next, dma, DI, ddt, dstage);
            dma_done = true;
            if (!ok) {          // a row sum reached the lazy limit: redo tile t exactly from its scores (still in s[P])
                exact_step(pc, std::integral_constant<int, 0>{}, false, kv0, q0w, cur.kv_lo, true);
                exact_step(pc, std::integral_constant<int, 1>{}, false, kv0, q0w, cur.kv_lo, true);
            }
            // P(t) in pf[P] waits for its PV, s[P ^ 1] = QK^T(t + 1)
        } else {
            if (dma) issue_tile(DI, ddt, dstage);
            dma_done = true;
            if (!skip) {
                if (pvp) pv_plain(pn, st_prev);
                if (!sv) qk_plain(pc, st_cur);
                exact_step(pc, std::integral_constant<int, 0>{}, need_mask, kv0, q0w, cur.kv_lo, false);
                exact_step(pc, std::integral_constant<int, 1>{}, need_mask, kv0, q0w, cur.kv_lo, false);
                pvp = true;
                if (next_act) {
                    qk_plain(pn, st_next);
                    sv = true;
                } else {
                    sv = false;
                }
            } else if (pvp) {       // (a wave past its causal extent with a pending tile: cannot happen — its last tile flushes below)
                pv_plain(pn, st_prev);
                pvp = false;
            }
        }
        if (dma) {
            ops_cur += 4;
            advance_dma();
        }
        // Q rows of the next item: V4_AHEAD - 1 iterations before this item ends (or at its first tile when it is shorter)
        if (nxt.valid && !q_inflight && (t >= cur.ntiles - V4_AHEAD || t == cur.t_lo) && t + V4_AHEAD >= cur.ntiles) {
            issue_q(nxt);
            ops_cur += QOPS;
            ops_since_q = 0;
            q_inflight = true;
        }
        bool finished = false;
        if (last) {
            if (!skip && pvp) pv_plain(pc, st_cur);          // P(t) of the item's last tile
            pvp = false;
            const int nst = store_item();
            ops_cur += nst;
            ops_since_q += nst;
        }
        // end of the iteration: tile g + 2 of the stream (issued at the top of the previous iteration) must have landed before
        // the barrier — everything issued after it may stay in flight; an item switch also needs the prefetched Q rows
        int allowed = ops_prev_after + ops_cur;
        if (last && q_inflight) allowed = min(allowed, ops_since_q);
        v4_wait_vm(allowed);
        __builtin_amdgcn_s_barrier();
        ops_prev_after = ops_cur - (dma ? 4 : 0);
        if (!(last && q_inflight)) ops_since_q += 0;
        ++g;
        if (last) {
            if (!nxt.valid) {
                finished = true;
            } else {
                cur = nxt;
                ++kc;
                decode(kc + 1, nxt);
                q_inflight = false;
                t = cur.t_lo;
                enter_item();
            }
        } else {
            ++t;
        }
        (void)dma_done;
        return finished;
    };

    while (true) {
        if (iteration(std::integral_constant<int, 0>{})) break;
        if (iteration(std::integral_constant<int, 1>{})) break;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores and any unused prefetch land before the LDS is released
}

// returns false when this kernel does not apply (the caller keeps attn_bf16_v2): head_dim 64, row-major V, whole kv tiles in
// the slab (kv_stride % 64 == 0), at least 256 query rows per item.
bool gar_attn_bf16_v4_try(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv, int hd, int q_row0,
                          int q_len, int q_total, int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev,
                          const int32_t* kv_start, int kv_prefix, hipStream_t s) {
    static const int enabled = [] {
        const char* e = getenv("GAR_ATTN_V4");
        return e ? atoi(e) : 1;
    }();
    if (!enabled || hd != 64 || q_len < 256 || (int64_t)kv_stride * 128 >= ((int64_t)1 << 31) || (int64_t)q_pad * 128 >= ((int64_t)1 << 31))
        return false;
    if ((kv_stride & 63) != 0 || Hq % Hkv != 0) return false;
    v4_args a;
    a.Q = (const bf16_t*)Q; a.K = (const bf16_t*)K; a.V = (const bf16_t*)V; a.O = (bf16_t*)O;
    a.kv_len_dev = kv_len_dev; a.kv_start = kv_start;
    a.B = B; a.Hq = Hq; a.Hkv = Hkv; a.q_row0 = q_row0; a.q_len = q_len; a.q_total = q_total; a.q_pad = q_pad;
    a.kv_len = kv_len; a.kv_stride = kv_stride; a.kv_prefix = kv_prefix;
    a.nqb = (q_len + 255) / 256;
    const int64_t n_items = (int64_t)B * Hq * a.nqb;
    if (n_items >= ((int64_t)1 << 30)) return false;
    static gar_once_per_device attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bf16_v4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bf16_v4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, V4_LDS);
    });
    const int cus = gar_num_cus();
    const int grid = (int)(n_items < cus ? n_items : cus);
    if (causal) hipLaunchKernelGGL(attn_bf16_v4_kernel<true>, dim3(grid), dim3(256), V4_LDS, s, a);
    else hipLaunchKernelGGL(attn_bf16_v4_kernel<false>, dim3(grid), dim3(256), V4_LDS, s, a);
    return true;
}
