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
__device__ __forceinline__ u32x2 v4_lds_b64(unsigned addr) {
    u32x2 d;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "i"(OFF));
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
    // a TAKEN branch costs a lone wave ~100 cycles of instruction refetch: the steady state (4: the newest tile DMA) falls through
    if (__builtin_expect(n >= 4 && n < 8, 1)) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        return;
    }
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
#ifdef V4_TIMELINE   /* diagnostic build (tools/attn_v4_timeline.py): shader-clock stamps between the segments of a tile iteration; workgroup 0
                        writes its per-wave sums over rows 0..3 of O when it is done */
#define V4_TL(i) { __builtin_amdgcn_sched_barrier(0); tl_t[i] = (unsigned)__builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); }
#define V4_TL_ACC { for (int i_ = 0; i_ < 6; ++i_) tl_sum[i_] += tl_t[i_ + 1] - tl_t[i_]; ++tl_n; }
#else
#define V4_TL(i)
#define V4_TL_ACC
#endif
// the lane index, opaque to the optimiser: per-lane addresses used once per ITEM are derived from it where they are used — hoisted
// to the kernel entry they are spilled around the tile loop, and every scratch reload is waited for with vmcnt(0), which drains the
// DMA ring at each item (the guide's pitfall: recompute per block)
__device__ __forceinline__ int v4_lane_opaque() {
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}


// Every matrix instruction below opens with `s_nop 1`: hipcc pads no hazards for an asm statement, and under register pressure
// (the causal instantiation) it parks an operand tuple — the -m start, a P fragment — elsewhere and copies it back with v_mov RIGHT
// in front of the statement: a VALU write followed at once by the MFMA's read of that register needs two wait states.
#if GAR_HALF_F16
#define V4_MFMA_OP "s_nop 1\n\tv_mfma_f32_32x32x16_f16"
#else
#define V4_MFMA_OP "s_nop 1\n\tv_mfma_f32_32x32x16_bf16"
#endif
// Matrix instructions as inline asm so that the register FILE of every operand is chosen here: with one wave per SIMD the
// compiler selects the AccVGPR form for every MFMA result, and the scores — which the VALU exponentiates — would come back
// through 64 v_accvgpr_read per tile. O (touched by MFMAs only), the Q fragments and the K / V fragments (ds_read straight into
// AccVGPRs) live in the accumulator file; scores, P and the -m start in arch VGPRs.
// hipcc does not pad hazards around asm: an MFMA result is read by compiler code only behind v4_drain() or a whole phase later.
__device__ __forceinline__ void v4_mfma_s_first(f32x16& d, const bf16x8& ka, const bf16x8& qb_, const f32x16& c) {
    asm volatile(V4_MFMA_OP " %0, %1, %2, %3" : "=&v"(d) : "a"(ka), "a"(qb_), "v"(c));
}
__device__ __forceinline__ void v4_mfma_s_acc(f32x16& d, const bf16x8& ka, const bf16x8& qb_) {
    asm volatile(V4_MFMA_OP " %0, %1, %2, %0" : "+v"(d) : "a"(ka), "a"(qb_));
}
__device__ __forceinline__ void v4_mfma_o_acc(f32x16& d, const bf16x8& va, const u32x4& pb) {
    asm volatile(V4_MFMA_OP " %0, %1, %2, %0" : "+a"(d) : "a"(va), "v"(pb));
}
__device__ __forceinline__ void v4_drain() { asm volatile("s_nop 15\n\ts_nop 7" ::: "memory"); }
template <int OFF>
__device__ __forceinline__ bf16x8 v4_lds_b128a(unsigned addr) {          // into AccVGPRs
    bf16x8 d;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "i"(OFF));
    return d;
}
template <int OFF>
__device__ __forceinline__ tr4_t v4_lds_tra(unsigned addr) {
    tr4_t d;
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=a"(d) : "v"(addr), "i"(OFF));
    return d;
}
template <int N>
__device__ __forceinline__ void v4_wait_lgkm_a(bf16x8& x) {
    asm volatile("s_waitcnt lgkmcnt(%1)" : "+a"(x) : "i"(N));
}
template <int N>
__device__ __forceinline__ void v4_wait_lgkm_a2(tr4_t& x, tr4_t& y) {
    asm volatile("s_waitcnt lgkmcnt(%2)" : "+a"(x), "+a"(y) : "i"(N));
}

// K / V fragments live in an asm-owned window of the accumulator file — a[192:223] = the eight K fragments of a tile (kd-major,
// kv block minor), a[224:255] = its eight V fragments ((kv block, 16-kv step)-major, d block minor; each two transposing 64-bit
// reads) — named literally in the asm strings and declared clobbered by every statement that writes them, so the compiler keeps
// nothing there: the two halves of a V fragment land in ONE 128-bit tuple (built from two asm outputs the compiler copies them
// together: 4 v_accvgpr_mov per fragment), and a fragment read one phase — or one loop iteration — ahead of its MFMA needs no
// compiler-visible live range across the barrier.
#define V4_KF0 192
#define V4_VF0 224
#define V4_FR_CLOBBER "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255"
#define V4_O_CLOBBER_ONLY "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
#define V4_QF_CLOBBER_ONLY "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95"
#define V4_OWNED V4_O_CLOBBER_ONLY, V4_QF_CLOBBER_ONLY, V4_FR_CLOBBER
#define V4_O_CLOBBER V4_OWNED
#define V4_QF_CLOBBER V4_OWNED
template <int F, int OFF>
__device__ __forceinline__ void v4_kfrag_read(unsigned addr) {          // K fragment F <- LDS
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%3" ::"v"(addr), "i"(V4_KF0 + 4 * F), "i"(V4_KF0 + 4 * F + 3), "i"(OFF)
                 : V4_OWNED);
}
template <int F, int HALF, int OFF>
__device__ __forceinline__ void v4_vfrag_read(unsigned addr) {          // half HALF (kv rows +0..3 / +4..7 of the step) of V fragment F
    asm volatile("ds_read_b64_tr_b16 a[%c1:%c2], %0 offset:%3" ::"v"(addr), "i"(V4_VF0 + 4 * F + 2 * HALF), "i"(V4_VF0 + 4 * F + 2 * HALF + 1),
                 "i"(OFF)
                 : V4_OWNED);
}
template <int N>
__device__ __forceinline__ void v4_wait_lgkm_n() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"i"(N) : V4_OWNED);
}
// O (a[0:63]: q-block qb, d block db at 32 qb + 16 db) and the Q fragments (a[64:95]: q-block qb, k-step kd at 64 + 16 qb + 4 kd) are
// asm-owned too: as "+a" operands the compiler kept O in arch VGPRs and copied it into the accumulator file around every asm
// block (70 v_accvgpr moves per phase). Compiler code reaches O only through v4_o_read / v4_o_write (item entry, re-base, epilogue).
#define V4_O0 0
#define V4_QF0 64
// every statement that touches an owned range declares ALL of them clobbered: the compiler then keeps nothing of its own in
// a[0:95] / a[192:255] across the kv loop (tests/test_abi.py audits the ISA: no compiler v_accvgpr_* on an owned register)
// S^T (arch VGPRs) = K fragment F x Q fragment (q-block QB, k-step F >> 1) (+ C)
template <int QB, int F>
__device__ __forceinline__ void v4_mfma_qk_first(f32x16& d, const f32x16& c) {
    asm volatile(V4_MFMA_OP " %0, a[%c2:%c3], a[%c4:%c5], %1"
                 : "=&v"(d)
                 : "v"(c), "i"(V4_KF0 + 4 * F), "i"(V4_KF0 + 4 * F + 3), "i"(V4_QF0 + 16 * QB + 4 * (F >> 1)),
                   "i"(V4_QF0 + 16 * QB + 4 * (F >> 1) + 3)
                 : V4_OWNED);
}
template <int QB, int F>
__device__ __forceinline__ void v4_mfma_qk_acc(f32x16& d) {
    asm volatile(V4_MFMA_OP " %0, a[%c1:%c2], a[%c3:%c4], %0"
                 : "+v"(d)
                 : "i"(V4_KF0 + 4 * F), "i"(V4_KF0 + 4 * F + 3), "i"(V4_QF0 + 16 * QB + 4 * (F >> 1)), "i"(V4_QF0 + 16 * QB + 4 * (F >> 1) + 3)
                 : V4_OWNED);
}
// O^T (q-block QB, d block F & 1) += V fragment F x P
template <int QB, int F>
__device__ __forceinline__ void v4_mfma_pv(const u32x4& pb) {
    asm volatile(V4_MFMA_OP " a[%c1:%c2], a[%c3:%c4], %0, a[%c1:%c2]" ::"v"(pb), "i"(V4_O0 + 32 * QB + 16 * (F & 1)),
                 "i"(V4_O0 + 32 * QB + 16 * (F & 1) + 15), "i"(V4_VF0 + 4 * F), "i"(V4_VF0 + 4 * F + 3)
                 : V4_O_CLOBBER);
}
// The half-bundles of the kv loop: [2 v_exp_f32 of the OTHER q-block's scores | one MFMA] as ONE statement. A lone wave issues about
// one instruction per 4-5 cycles whatever its type, so a 32-cycle MFMA shadows ~5 issue slots (the guide's budget) — a separate
// `s_nop` in front of every MFMA would take one of them; here the two exponentials the slice needs anyway are the wait states
// between a compiler copy of an operand and the MFMA's read of it.
#if GAR_HALF_F16
#define V4_MFMA_RAW "v_mfma_f32_32x32x16_f16"
#else
#define V4_MFMA_RAW "v_mfma_f32_32x32x16_bf16"
#endif
template <int QB, int F>
__device__ __forceinline__ void v4_hb_qk_first(f32x16& d, const f32x16& c, float& p0, float& p1, float s0, float s1) {
    asm volatile("v_exp_f32 %1, %4\n\tv_exp_f32 %2, %5\n\t" V4_MFMA_RAW " %0, a[%c6:%c7], a[%c8:%c9], %3"
                 : "=&v"(d), "=&v"(p0), "=&v"(p1)
                 : "v"(c), "v"(s0), "v"(s1), "i"(V4_KF0 + 4 * F), "i"(V4_KF0 + 4 * F + 3), "i"(V4_QF0 + 16 * QB + 4 * (F >> 1)),
                   "i"(V4_QF0 + 16 * QB + 4 * (F >> 1) + 3)
                 : V4_OWNED);
}
template <int QB, int F>
__device__ __forceinline__ void v4_hb_qk_acc(f32x16& d, float& p0, float& p1, float s0, float s1) {
    asm volatile("v_exp_f32 %1, %3\n\tv_exp_f32 %2, %4\n\t" V4_MFMA_RAW " %0, a[%c5:%c6], a[%c7:%c8], %0"
                 : "+v"(d), "=&v"(p0), "=&v"(p1)
                 : "v"(s0), "v"(s1), "i"(V4_KF0 + 4 * F), "i"(V4_KF0 + 4 * F + 3), "i"(V4_QF0 + 16 * QB + 4 * (F >> 1)),
                   "i"(V4_QF0 + 16 * QB + 4 * (F >> 1) + 3)
                 : V4_OWNED);
}
template <int QB, int F>
__device__ __forceinline__ void v4_hb_pv(const u32x4& pb, float& p0, float& p1, float s0, float s1) {
    asm volatile("v_exp_f32 %0, %3\n\tv_exp_f32 %1, %4\n\t" V4_MFMA_RAW " a[%c5:%c6], a[%c7:%c8], %2, a[%c5:%c6]"
                 : "=&v"(p0), "=&v"(p1)
                 : "v"(pb), "v"(s0), "v"(s1), "i"(V4_O0 + 32 * QB + 16 * (F & 1)), "i"(V4_O0 + 32 * QB + 16 * (F & 1) + 15),
                   "i"(V4_VF0 + 4 * F), "i"(V4_VF0 + 4 * F + 3)
                 : V4_O_CLOBBER);
}
template <int R>
__device__ __forceinline__ float v4_o_read() {
    float x;
    asm volatile("v_accvgpr_read_b32 %0, a%c1" : "=v"(x) : "i"(V4_O0 + R) : V4_OWNED);
    return x;
}
template <int R>
__device__ __forceinline__ void v4_o_write(float x) {
    asm volatile("v_accvgpr_write_b32 a%c1, %0" ::"v"(x), "i"(V4_O0 + R) : V4_O_CLOBBER);
}
template <int I, int OFF>
__device__ __forceinline__ void v4_qfrag_read(unsigned addr) {          // Q fragment I (= 4 qb + kd) <- LDS
    asm volatile("ds_read_b128 a[%c1:%c2], %0 offset:%3" ::"v"(addr), "i"(V4_QF0 + 4 * I), "i"(V4_QF0 + 4 * I + 3), "i"(OFF) : V4_QF_CLOBBER);
}
template <bool CAUSAL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void attn_bf16_v4_kernel(const v4_args a) {
#if defined(__HIP_DEVICE_COMPILE__)      // gfx950 inline asm (AccVGPR constraints): the host pass only needs the launch stub
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int PFX = CAUSAL ? 0 : a.kv_prefix;
    const int kv_len = (a.kv_len_dev ? ((const __attribute__((address_space(4))) int32_t*)a.kv_len_dev)[0] : a.kv_len) - PFX;
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
    auto decode = [&](int k, int& b, int& head, int& qb) __attribute__((always_inline)) -> bool {
        const int L = blockIdx.x;
        const int it = (G & 7) == 0 ? k * G + (L & 7) * (G >> 3) + (L >> 3) : k * G + L;
        const bool valid = it < n_items;
        const int itc = valid ? it : 0;
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
        return valid;
    };
    auto kv_lo_of = [&](int b) __attribute__((always_inline)) -> int {
        // through the constant address space: a scalar load (a vector load here would be waited for with vmcnt(0) and drain the DMA ring)
        return a.kv_start ? max(min(((const __attribute__((address_space(4))) int32_t*)a.kv_start)[b], kv_len - 1), 0) : 0;
    };
    auto ntiles_of = [&](int qb, int kv_lo) __attribute__((always_inline)) -> int {
        int kv_end = kv_len;
        if (CAUSAL) kv_end = min(kv_len, max(min(a.q_row0 + qb * 256 + 255, q_end - 1) + coff, kv_lo) + 1);
        return (kv_end + 63) >> 6;
    };
    auto slab_rsrc = [&](const bf16_t* base, int b, int head) __attribute__((always_inline)) -> __amdgpu_buffer_rsrc_t {
        const bf16_t* p = base + ((int64_t)b * a.Hkv + head / gsz) * (int64_t)a.kv_stride * 64;
        return __builtin_amdgcn_make_buffer_rsrc((void*)p, 0, (int)slab, 0x00020000);
    };

    // ---- LDS-DMA: a 64-row tile = 8 pieces of 8 rows x 128 B (1 KiB, lane-linear in LDS); wave w brings pieces w and 4 + w of K
    // and of V. The XOR swizzle of the 16-byte chunks is applied to the SOURCE offset (K: key (row >> 1) & 7 -> conflict-free
    // b128 fragment reads; V: key 4 ((row >> 1) & 1) -> the four rows of a transposing-read block cover all banks).
    const int drow = wave * 8 + (lane >> 3);
    const int voffK = drow * 128 + (((lane & 7) ^ ((drow >> 1) & 7)) << 4);
    const int voffV = drow * 128 + (((lane & 7) ^ (((drow >> 1) & 1) << 2)) << 4);
    // DMA cursor: tile d_t of the kd-th item of this workgroup
    int kdi = 0, d_t, d_nt;
    bool d_valid;
    __amdgpu_buffer_rsrc_t d_rsK, d_rsV;
    auto dma_enter = [&](int k) __attribute__((always_inline)) {
        int b, head, qb;
        (void)decode(k, b, head, qb);         // past the last item: item 0 again — the ring keeps its cadence (no branch in the tile
        d_valid = true;                       // loop, a constant vmcnt), the extra tiles are never read
        const int lo = kv_lo_of(b);
        d_t = lo >> 6;
        d_nt = ntiles_of(qb, lo);
        d_rsK = slab_rsrc(a.K, b, head);
        d_rsV = slab_rsrc(a.V, b, head);
    };
    auto dma_advance = [&]() __attribute__((always_inline)) {
        if (++d_t >= d_nt) dma_enter(++kdi);
    };
    auto dma_piece = [&](auto ic, int stage) __attribute__((always_inline)) {        // piece i of the cursor's tile: K K V V
        constexpr int i = decltype(ic)::value;
#ifdef V4_KO_DMA
        return;
#endif
        char* ks = smem + stage * V4_STAGE;
        const int base = (d_t * 64 + PFX) * 128;
        if (i < 2)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsK, LDS_AS(ks + ((i & 1) * 4 + wave) * 1024), 16, voffK + base + (i & 1) * 4096, 0, 0, 0);
        else
            __builtin_amdgcn_raw_ptr_buffer_load_lds(d_rsV, LDS_AS(ks + 8192 + ((i & 1) * 4 + wave) * 1024), 16,
                                                     voffV + base + (i & 1) * 4096, 0, 0, 0);
    };
    auto dma_tile = [&](int stage) __attribute__((always_inline)) {
        v4_for<0, 4>([&](auto ic) __attribute__((always_inline)) { dma_piece(ic, stage); });
    };
    // the wave's own 64 Q rows of an item (row-major image, no swizzle: read once per item) + the prefix key / value row
    char* const qbuf = smem + V4_QBUF + wave * V4_QW;
    const unsigned qbuf_a = lds0 + V4_QBUF + wave * V4_QW;
    const int voffQ = (lane >> 3) * 128 + ((lane & 7) << 4);
    const int QOPS = 8 + (PFX ? 2 : 0);
    auto issue_q = [&](int k) __attribute__((always_inline)) {
#ifdef V4_KO_Q
        if (k > 0) return;
#endif
        int b, head, qb;
        (void)decode(k, b, head, qb);
        const bf16_t* Qp = a.Q + ((int64_t)b * a.Hq + head) * (int64_t)a.q_pad * 64;
        const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc((void*)Qp, 0, a.q_pad * 128, 0x00020000);
        const int row0 = a.q_row0 + qb * 256 + wave * 64;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, LDS_AS(qbuf + p * 1024), 16, voffQ + (row0 + p * 8) * 128, 0, 0, 0);
        if (PFX) {          // row 0 of the slab, 8 chunks (replicated over the 8 row slots of the piece)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(slab_rsrc(a.K, b, head), LDS_AS(qbuf + 8192), 16, (lane & 7) << 4, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(slab_rsrc(a.V, b, head), LDS_AS(qbuf + 9216), 16, (lane & 7) << 4, 0, 0, 0);
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

    // ---- wave state of the current item. ONE set of scores and ONE set of P: the two q-blocks run half an iteration apart
    //   phase A(j): MFMA { S^T(j, qb1) = K(j) Q1^T ;  O1^T += V^T(j-1) P1(j-1) }   |  VALU softmax(j, qb0)   | LDS: K(j+1), V(j) fragments
    //   phase B(j): MFMA { S^T(j+1, qb0) = K(j+1) Q0^T ;  O0^T += V^T(j) P0(j) }   |  VALU softmax(j, qb1)   | the tile DMA
    // so every score / P register is consumed before the matrix pipe rewrites it, a q-block's O is complete up to the
    // previous tile whenever its running max may have to be re-based, and the K / V fragments of a tile — read ONCE into the
    // AccVGPR window, each slot refilled right behind its last MFMA of phase A — serve both q-blocks.
    f32x16 negm[2];                  // -m of the q-block's row: the C operand of the first QK^T MFMA
    f32x16 s[2][2];                  // [q-block][kv block]
    u32x4 pf[2][2][2];               // [q-block][kv block][16-kv step]: P as PV B operands
    float m_run[2], l_run[2];
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4 = {0u, 0u, 0u, 0u};
    const auto Q0 = std::integral_constant<int, 0>{};
    const auto Q1 = std::integral_constant<int, 1>{};

    // fragment f of a tile: K (kd = f >> 1, kv block f & 1); V ((kv block, 16-kv step) = f >> 1, d block f & 1)
    auto k_read = [&](auto fc, const unsigned (&ka)[4]) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value;
#ifndef V4_KO_LDS
        v4_kfrag_read<f, (f & 1) * 4096>(ka[f >> 1]);
#endif
    };
    auto v_read = [&](auto fc, const unsigned (&va)[2]) __attribute__((always_inline)) {
        constexpr int f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1, db = f & 1;
#ifndef V4_KO_LDS
        v4_vfrag_read<f, 0, (kb * 32 + tt * 16) * 128>(va[db]);
        v4_vfrag_read<f, 1, (kb * 32 + tt * 16 + 4) * 128>(va[db]);
#endif
    };
    auto qk_mfma = [&](auto qc, auto fc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qc)::value, f = decltype(fc)::value, kd = f >> 1, kb = f & 1;
#ifndef V4_KO_MFMA
        if constexpr (kd == 0) v4_mfma_qk_first<qb, f>(s[qb][kb], negm[qb]);
        else v4_mfma_qk_acc<qb, f>(s[qb][kb]);
#endif
    };
    auto pv_mfma = [&](auto qc, auto fc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qc)::value, f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1, db = f & 1;
#ifndef V4_KO_MFMA
        v4_mfma_pv<qb, f>(pf[qb][kb][tt]);
#endif
    };
    // p = exp2(s) (s carries -m), row sum, bf16 pack of q-block qb -> pf[qb]; returns this lane's partial row sum
    auto exp_pack = [&](auto qc) __attribute__((always_inline)) -> float {
        constexpr int qb = decltype(qc)::value;
        float ps0 = 0.f, ps1 = 0.f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            float p[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(s[qb][kb][r]);
#pragma unroll
            for (int r = 0; r < 16; r += 2) { ps0 += p[r]; ps1 += p[r + 1]; }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                pf[qb][kb][tt] = u32x4{pack_bf2(p[tt * 8 + 0], p[tt * 8 + 1]), pack_bf2(p[tt * 8 + 2], p[tt * 8 + 3]),
                                       pack_bf2(p[tt * 8 + 4], p[tt * 8 + 5]), pack_bf2(p[tt * 8 + 6], p[tt * 8 + 7])};
        }
        return ps0 + ps1;
    };
    // exact softmax preparation of q-block qb's scores (first tile, masked tiles, lazy-max overflow): optional mask, exact tile
    // max, re-base of the running max where it grew — O (complete up to the previous tile by construction of the schedule), l,
    // the scores, the -m start. The lazy exponentiation that follows is then exact.
    auto exact_prepare = [&](auto qc, bool need_mask, bool scale_o, int kv0, int q0w, int kv_lo) __attribute__((always_inline)) {
        constexpr int qb = decltype(qc)::value;
        if (need_mask) {
            const int qi = q0w + qb * 32 + l31;
            const int lim = CAUSAL ? min(kv_len - 1, max(qi + coff, kv_lo)) : kv_len - 1;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int kv = kv0 + kb * 32 + ((r >> 3) << 4) + h * 8 + (r & 7);
                    s[qb][kb][r] = (kv <= lim && kv >= kv_lo) ? s[qb][kb][r] : -INFINITY;
                }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[qb][kb][r]);
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
            if (scale_o) {          // O of this q-block through the arch file (drained MFMAs on both sides: the callers' v4_drain)
                v4_drain();
                v4_for<0, 32>([&](auto rc) __attribute__((always_inline)) {
                    constexpr int r = decltype(rc)::value;
                    v4_o_write<32 * qb + r>(v4_o_read<32 * qb + r>() * alpha);
                });
            }
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) s[qb][kb][r] += shift;
#pragma unroll
            for (int r = 0; r < 16; ++r) negm[qb][r] = -m_nu;
        }
    };
    // softmax half-slice m (0..15) of q-block qb: elements 2m, 2m+1 of the 32 scores of this lane — 2 exp | 2 add, 1 cvt_pk. With one
    // wave per SIMD nothing covers the latency of a result but the wave's own later instructions: the exponentials of slice m issue
    // in half-bundle m, their sums and pack in half-bundle m + 1.
    float ps[2][2], pe[2][2];
    auto sm_exp = [&](auto qc, auto mc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qc)::value, m = decltype(mc)::value, kb = m >> 3, r0 = (m & 7) * 2;
#ifdef V4_KO_SM
        return;
#endif
        pe[m & 1][0] = __builtin_amdgcn_exp2f(s[qb][kb][r0]);
        pe[m & 1][1] = __builtin_amdgcn_exp2f(s[qb][kb][r0 + 1]);
    };
    auto sm_use = [&](auto qc, auto mc) __attribute__((always_inline)) {
        constexpr int qb = decltype(qc)::value, m = decltype(mc)::value, kb = m >> 3, r0 = (m & 7) * 2;
#ifdef V4_KO_SM
        return;
#endif
        ps[qb][0] += pe[m & 1][0];
        ps[qb][1] += pe[m & 1][1];
        pf[qb][kb][r0 >> 3][(r0 & 7) >> 1] = pack_bf2(pe[m & 1][0], pe[m & 1][1]);
        asm volatile("" : "+v"(pf[qb][kb][r0 >> 3]), "+v"(ps[qb][0]), "+v"(ps[qb][1]));
    };
    // half-bundle m of a phase: the MFMA of q-block QM (QK^T fragment f of q-block QM, or its PV fragment f) with the exponentials of
    // slice m of q-block QS inside the same statement; the sums / pack of slice m - 1 behind it
    auto hb_qk = [&](auto qmc, auto fc, auto qsc, auto mc) __attribute__((always_inline)) {
        constexpr int qm = decltype(qmc)::value, f = decltype(fc)::value, kd = f >> 1, kb = f & 1;
        constexpr int qs = decltype(qsc)::value, m = decltype(mc)::value, skb = m >> 3, r0 = (m & 7) * 2;
#if defined(V4_KO_SM) || defined(V4_KO_MFMA)
        qk_mfma(qmc, fc);
        sm_exp(qsc, mc);
#else
        if constexpr (kd == 0) v4_hb_qk_first<qm, f>(s[qm][kb], negm[qm], pe[m & 1][0], pe[m & 1][1], s[qs][skb][r0], s[qs][skb][r0 + 1]);
        else v4_hb_qk_acc<qm, f>(s[qm][kb], pe[m & 1][0], pe[m & 1][1], s[qs][skb][r0], s[qs][skb][r0 + 1]);
#endif
        if constexpr (m > 0) sm_use(qsc, std::integral_constant<int, (m > 0 ? m - 1 : 0)>{});
    };
    auto hb_pv = [&](auto qmc, auto fc, auto qsc, auto mc) __attribute__((always_inline)) {
        constexpr int qm = decltype(qmc)::value, f = decltype(fc)::value, st = f >> 1, kb = st >> 1, tt = st & 1;
        constexpr int qs = decltype(qsc)::value, m = decltype(mc)::value, skb = m >> 3, r0 = (m & 7) * 2;
#if defined(V4_KO_SM) || defined(V4_KO_MFMA)
        pv_mfma(qmc, fc);
        sm_exp(qsc, mc);
#else
        v4_hb_pv<qm, f>(pf[qm][kb][tt], pe[m & 1][0], pe[m & 1][1], s[qs][skb][r0], s[qs][skb][r0 + 1]);
#endif
        if constexpr (m > 0) sm_use(qsc, std::integral_constant<int, (m > 0 ? m - 1 : 0)>{});
    };
    const float lazy_lim = (float)(1u << H16_MAX_LOG2);

    // the V fragment window multiplies a zero P in the first tile of the first item: no NaN bit patterns in it
    asm volatile("v_accvgpr_write_b32 a224, 0\n\tv_accvgpr_write_b32 a225, 0\n\tv_accvgpr_write_b32 a226, 0\n\tv_accvgpr_write_b32 a227, 0\n\t"
                 "v_accvgpr_write_b32 a228, 0\n\tv_accvgpr_write_b32 a229, 0\n\tv_accvgpr_write_b32 a230, 0\n\tv_accvgpr_write_b32 a231, 0\n\t"
                 "v_accvgpr_write_b32 a232, 0\n\tv_accvgpr_write_b32 a233, 0\n\tv_accvgpr_write_b32 a234, 0\n\tv_accvgpr_write_b32 a235, 0\n\t"
                 "v_accvgpr_write_b32 a236, 0\n\tv_accvgpr_write_b32 a237, 0\n\tv_accvgpr_write_b32 a238, 0\n\tv_accvgpr_write_b32 a239, 0\n\t"
                 "v_accvgpr_write_b32 a240, 0\n\tv_accvgpr_write_b32 a241, 0\n\tv_accvgpr_write_b32 a242, 0\n\tv_accvgpr_write_b32 a243, 0\n\t"
                 "v_accvgpr_write_b32 a244, 0\n\tv_accvgpr_write_b32 a245, 0\n\tv_accvgpr_write_b32 a246, 0\n\tv_accvgpr_write_b32 a247, 0\n\t"
                 "v_accvgpr_write_b32 a248, 0\n\tv_accvgpr_write_b32 a249, 0\n\tv_accvgpr_write_b32 a250, 0\n\tv_accvgpr_write_b32 a251, 0\n\t"
                 "v_accvgpr_write_b32 a252, 0\n\tv_accvgpr_write_b32 a253, 0\n\tv_accvgpr_write_b32 a254, 0\n\tv_accvgpr_write_b32 a255, 0"
                 ::: V4_OWNED);

    // ---- prologue: the first V4_AHEAD tiles of this workgroup's stream and the first item's Q rows
    int c_b, c_head, c_qb;
    if (!decode(0, c_b, c_head, c_qb)) return;
    dma_enter(0);
    int g = 0;                              // position in the workgroup's tile stream modulo V4_NS: tile g lives in stage g
    auto ring = [](int gs, int k) { const int x = gs + k; return x >= V4_NS ? x - V4_NS : x; };
#pragma unroll
    for (int i = 0; i < V4_AHEAD; ++i) {
        if (d_valid) {
            dma_tile(i % V4_NS);
            dma_advance();
        }
    }
    issue_q(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
#ifdef V4_TIMELINE
    unsigned tl_t[8] = {}, tl_sum[8] = {}, tl_n = 0, tl_items = 0;
    const unsigned tl_begin = (unsigned)__builtin_amdgcn_s_memtime();
#endif
    int ops_prev_after = 0;                 // VMEM operations issued in the previous iteration after its tile DMA
    int ops_since_q = 0;                    // ... issued after the newest Q prefetch

    for (int kc = 0;; ++kc) {
        // ---- item entry: Q fragments out of the wave's LDS rows (prefetched), the folded prefix key / value as the initial state
        const int kv_lo = kv_lo_of(c_b);
        const int t_lo = kv_lo >> 6, nt = ntiles_of(c_qb, kv_lo);
        const int q0w = a.q_row0 + c_qb * 256 + wave * 64;
        const bool wave_active = q0w < q_end;
        int n_b, n_head, n_qb;
        const bool n_valid = decode(kc + 1, n_b, n_head, n_qb);
        v4_wait_vm(ops_since_q);            // the Q rows of this item have landed (a no-op unless the previous item was very short)
        const int lane_i = v4_lane_opaque(), l31_i = lane_i & 31, h_i = lane_i >> 5;
        const unsigned qrow_a = qbuf_a + l31_i * 128 + (h_i << 4);          // this lane's chunk h of Q row l31 (q-block 1: + 4096)
        v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
            constexpr int f = decltype(fc)::value, qb = f >> 2, kd = f & 3;
            v4_qfrag_read<f, qb * 4096 + kd * 32>(qrow_a);
        });
#pragma unroll
        for (int qb = 0; qb < 2; ++qb) {
            m_run[qb] = -INFINITY;
            l_run[qb] = 0.f;
            negm[qb] = zero16;
        }
        if (!PFX) v4_for<0, 64>([&](auto rc) __attribute__((always_inline)) { v4_o_write<decltype(rc)::value>(0.f); });
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) pf[1][kb][tt] = zero4;      // "P of the tile before the first" of q-block 1: its PV adds nothing
        if (PFX) {
            // s0 = q . k0 (q carries scale * log2e): this lane holds dims 16 kd + 8 h .. + 8 of its query rows. LDS reads as asm: a
            // compiler-visible read behind the LDS-DMA is waited for with vmcnt(0), which would drain the tile ring at every item
            bf16x8 qv[2][4], k0f[4];
            u32x2 v0f[2][4];
            float ov[2][2][16];
            v4_for<0, 4>([&](auto kc_) __attribute__((always_inline)) {
                constexpr int kd = decltype(kc_)::value;
                qv[0][kd] = v4_lds_b128<kd * 32>(qrow_a);
                qv[1][kd] = v4_lds_b128<4096 + kd * 32>(qrow_a);
                k0f[kd] = v4_lds_b128<8192 + kd * 32>(qbuf_a + (h_i << 4));
                v0f[0][kd] = v4_lds_b64<9216 + kd * 16>(qbuf_a + h_i * 8);
                v0f[1][kd] = v4_lds_b64<9216 + 64 + kd * 16>(qbuf_a + h_i * 8);
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            V4_FENCE;
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                float part = 0.f;
#pragma unroll
                for (int kd = 0; kd < 4; ++kd)
#pragma unroll
                    for (int e = 0; e < 8; ++e) part = __builtin_fmaf(bf2f((bf16_t)qv[qb][kd][e]), bf2f((bf16_t)k0f[kd][e]), part);
                m_run[qb] = part + __shfl_xor(part, 32, 64);
                l_run[qb] = h_i == 0 ? 1.0f : 0.f;           // the two halves' l are added at the end
#pragma unroll
                for (int r = 0; r < 16; ++r) negm[qb][r] = -m_run[qb];
                // O0 = v0: register r of d-block d is d index 32 d + (r & 3) + 8 (r >> 2) + 4 h
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        ov[qb][d][gq * 4 + 0] = unpk_lo(v0f[d][gq][0]);
                        ov[qb][d][gq * 4 + 1] = unpk_hi(v0f[d][gq][0]);
                        ov[qb][d][gq * 4 + 2] = unpk_lo(v0f[d][gq][1]);
                        ov[qb][d][gq * 4 + 3] = unpk_hi(v0f[d][gq][1]);
                    }
            }
            v4_for<0, 64>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                v4_o_write<r>(ov[r >> 5][(r >> 4) & 1][r & 15]);
            });
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the Q rows are consumed: the next prefetch may overwrite them
        bool q_issued = false;
        if (wave_active) {
            // "phase B(-1)": the K fragments of the item's first tile and q-block 0's scores of it
            unsigned ka[4];
#pragma unroll
            for (int kd = 0; kd < 4; ++kd) ka[kd] = kaddr0[kd] + g * V4_STAGE;
            v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) { k_read(fc, ka); });
            v4_wait_lgkm_n<0>();
            v4_drain();                     // negm (VALU) -> MFMA C operand
            v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) { qk_mfma(Q0, fc); });
            v4_drain();
        }

        // one tile of the item. STEADY (compile-time): a tile every wave of the workgroup takes unmasked and lazily, that is neither the
        // item's first nor among its last V4_AHEAD — no flag, no cursor decision, a constant vmcnt: the hot loop's scalar code and
        // branches vanish (a lone wave pays ~100 cycles of refetch per taken branch and ~4 cycles per instruction of any kind)
        auto tile_iter = [&](auto steady_c, int t) __attribute__((always_inline)) -> bool {
            constexpr bool ST = decltype(steady_c)::value;
            V4_TL(0)
            const int kv0 = t * 64;
            const bool first = !ST && t == t_lo, last = !ST && t + 1 >= nt;
            const int st_cur = g, st_next = ring(g, 1), dstage = ring(g, V4_AHEAD);
            const bool act = ST ? wave_active : wave_active && !(CAUSAL && kv0 > max(q0w + 63 + coff, kv_lo));
            const bool next_act = ST ? act : act && !last && !(CAUSAL && kv0 + 64 > max(q0w + 63 + coff, kv_lo));
            const bool need_mask = !ST && ((kv0 + 64 > kv_len) || kv0 < kv_lo || (CAUSAL && kv0 + 63 > q0w + coff));
            const bool exact = first || need_mask;
            const bool have_o = !first || PFX;
            if (__builtin_expect(act, 1)) {
                if (__builtin_expect(exact, 0)) {
                    exact_prepare(Q0, need_mask, have_o, kv0, q0w, kv_lo);
                    v4_drain();
                }
                // ---- phase A
                {
                    unsigned ka[4], va[2];
#pragma unroll
                    for (int kd = 0; kd < 4; ++kd) ka[kd] = kaddr0[kd] + st_next * V4_STAGE;
                    va[0] = vaddr0[0] + (unsigned)(st_cur * V4_STAGE);
                    va[1] = vaddr0[1] + (unsigned)(st_cur * V4_STAGE);
                    ps[0][0] = ps[0][1] = 0.f;
                    v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
                        constexpr int f = decltype(fc)::value;
                        V4_FENCE;
                        hb_qk(Q1, fc, Q0, std::integral_constant<int, 2 * f>{});       // S^T(t, q-block 1), K fragment f of tile t ...
                        k_read(fc, ka);                          // ... whose slot takes fragment f of tile t + 1
                        V4_FENCE;
                        hb_pv(Q1, fc, Q0, std::integral_constant<int, 2 * f + 1>{});   // O1^T += V^T(t-1) P1(t-1), V fragment f of tile t - 1 ...
                        v_read(fc, va);                          // ... whose slot takes fragment f of tile t
                    });
                    V4_FENCE;
                    sm_use(Q0, std::integral_constant<int, 15>{});
                }
                V4_TL(1)
                {
                    const float t0 = ps[0][0] + ps[0][1];
                    if (__builtin_expect(__all(t0 < lazy_lim), 1)) {
                        l_run[0] += t0;
                    } else {            // a row sum reached the lazy limit: q-block 0 again, exactly (its scores are intact until phase B)
                        exact_prepare(Q0, false, true, kv0, q0w, kv_lo);
                        l_run[0] += exp_pack(Q0);
                        v4_drain();
                    }
                }
                if (__builtin_expect(exact, 0)) {
                    v4_drain();
                    exact_prepare(Q1, need_mask, have_o, kv0, q0w, kv_lo);
                    v4_drain();
                }
                V4_TL(2)
                // ---- phase B
                {
                    ps[1][0] = ps[1][1] = 0.f;
                    v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) {
                        constexpr int f = decltype(fc)::value;
                        V4_FENCE;
                        // the 24 fragment reads of phase A went out in the order K0 V0 V0' K1 V1 V1' ...: fragment pair f is read 3 f + 3
#ifdef V4_SAFE_WAIT
                        v4_wait_lgkm_n<0>();
#else
                        v4_wait_lgkm_n<(21 - 3 * f > 15) ? 15 : 21 - 3 * f>();
#endif
                        hb_qk(Q0, fc, Q1, std::integral_constant<int, 2 * f>{});       // S^T(t+1, q-block 0)
                        V4_FENCE;
                        hb_pv(Q0, fc, Q1, std::integral_constant<int, 2 * f + 1>{});   // O0^T += V^T(t) P0(t)
                        if constexpr (f < 4) dma_piece(fc, dstage);
                    });
                    V4_FENCE;
                    sm_use(Q1, std::integral_constant<int, 15>{});
                }
                V4_TL(3)
                {
                    const float t1 = ps[1][0] + ps[1][1];
                    if (__builtin_expect(__all(t1 < lazy_lim), 1)) {
                        l_run[1] += t1;
                    } else {
                        exact_prepare(Q1, false, true, kv0, q0w, kv_lo);
                        l_run[1] += exp_pack(Q1);
                        v4_drain();
                    }
                }
                if (__builtin_expect(!next_act, 0)) {    // this wave's last tile of the item: q-block 1's PV of it (the V fragments of tile t are in the window)
                    v4_for<0, 8>([&](auto fc) __attribute__((always_inline)) { pv_mfma(Q1, fc); });
                    v4_drain();
                }
            } else {
                dma_tile(dstage);
            }
            if (ST) {
                dma_advance();
                ops_since_q += 4;
                V4_TL(4)
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");          // tile g + 2 landed; the newest tile DMA stays in flight
                V4_TL(5)
                __builtin_amdgcn_s_barrier();
                V4_TL(6)
                V4_TL_ACC
                g = ring(g, 1);
                return false;
            }
            int ops_cur = 4;
            dma_advance();
            // Q rows of the next item: fetched V4_AHEAD - 1 tiles before this item ends (at its first tile when it is shorter)
            if (__builtin_expect(n_valid && !q_issued && t + V4_AHEAD >= nt, 0)) {
                issue_q(kc + 1);
                ops_cur += QOPS;
                ops_since_q = 0;
                q_issued = true;
            } else {
                ops_since_q += 4;
            }
            // tile g + 2 of the stream (issued at the top of the previous iteration) must have landed before the barrier:
            // everything issued after it may stay in flight
            V4_TL(4)
            v4_wait_vm(ops_prev_after + ops_cur);
            V4_TL(5)
            __builtin_amdgcn_s_barrier();
            V4_TL(6)
            V4_TL_ACC
            ops_prev_after = ops_cur - 4;
            g = ring(g, 1);
            return last;
        };
        // tiles t_lo < t < t_hi are STEADY for every wave: unmasked for the block's first row (causal), whole (kv tail), and followed by
        // at least V4_AHEAD more tiles of the item (the DMA cursor stays inside it, the next item's Q rows are not due yet)
        int t_hi = min(nt - V4_AHEAD, kv_len >> 6);
        if (CAUSAL) t_hi = min(t_hi, (a.q_row0 + c_qb * 256 + coff + 1) >> 6);
        for (int t = t_lo;;) {
            if (t > t_lo && t < t_hi && ops_prev_after == 0) {
                do {
                    tile_iter(std::true_type{}, t);
                    ++t;
                } while (t < t_hi);
                continue;
            }
            if (tile_iter(std::false_type{}, t)) break;
            ++t;
        }

        // ---- epilogue of the item: O^T fragments -> rows through the wave's LDS piece -> 16-byte stores (8 rows x 128 B each)
        if (wave_active) {
            const int lane_e = v4_lane_opaque(), l31_e = lane_e & 31, h_e = lane_e >> 5;
            char* const obuf_e = smem + V4_OBUF + wave * 8192;
            float oe[2][2][16];
            v4_drain();
            v4_for<0, 64>([&](auto rc) __attribute__((always_inline)) {
                constexpr int r = decltype(rc)::value;
                oe[r >> 5][(r >> 4) & 1][r & 15] = v4_o_read<r>();
            });
#pragma unroll
            for (int qb = 0; qb < 2; ++qb) {
                const float l_tot = l_run[qb] + __shfl_xor(l_run[qb], 32, 64);
                const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
                const int row = qb * 32 + l31_e;
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int gq = 0; gq < 4; ++gq)
                        *reinterpret_cast<u32x2*>(obuf_e + row * 128 + (((d * 4 + gq) ^ (row & 7)) << 4) + h_e * 8) =
                            u32x2{pack_bf2(oe[qb][d][gq * 4 + 0] * inv, oe[qb][d][gq * 4 + 1] * inv),
                                  pack_bf2(oe[qb][d][gq * 4 + 2] * inv, oe[qb][d][gq * 4 + 3] * inv)};
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const int nst = min(8, (q_end - q0w + 7) >> 3);           // wave-uniform
            bf16_t* const Ow = a.O + ((int64_t)c_b * a.q_total + q0w) * ((int64_t)a.Hq * 64) + c_head * 64;
            const int rr = lane_e >> 3, c = lane_e & 7;
            const unsigned o_lane = (unsigned)rr * (unsigned)(a.Hq * 128) + (unsigned)c * 16u;      // byte offset of (row rr, chunk c)
            for (int i = 0; i < nst; ++i) {
                const int row = i * 8 + rr;
                const u32x4 v = *reinterpret_cast<const u32x4*>(obuf_e + row * 128 + ((c ^ (row & 7)) << 4));
#ifndef V4_KO_STORE
                if (q0w + row < q_end)
#else
                if (q0w + row < q_end && a.q_len < 0)
#endif
                    *reinterpret_cast<u32x4*>((char*)Ow + (size_t)i * 8u * (size_t)(a.Hq * 128) + o_lane) = v;
            }
            ops_prev_after += nst;
            ops_since_q += nst;
        }
#ifdef V4_TIMELINE
        ++tl_items;
#endif
        if (!n_valid) break;
        c_b = n_b;
        c_head = n_head;
        c_qb = n_qb;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // stores and any unused prefetch land before the LDS is released
#ifdef V4_TIMELINE
    if (blockIdx.x == 0 && lane < 16) {
        const unsigned tl_total = (unsigned)__builtin_amdgcn_s_memtime() - tl_begin;
        unsigned v = 0u;
        for (int i = 0; i < 6; ++i) v = lane == i ? tl_sum[i] : v;
        v = lane == 6 ? tl_n : lane == 7 ? tl_total : lane == 8 ? tl_items : v;
        reinterpret_cast<unsigned*>(a.O + (int64_t)wave * ((int64_t)a.Hq * 64))[lane] = v;
    }
#endif
#endif
}

// Diagnostic build only since round 6 (tools/attn_v4/README.md): the variant library takes every shape the kernel is built for.
// returns false when this kernel does not apply (the caller keeps attn_bf16_v2): head_dim 64, row-major V, whole kv tiles in
// the slab (kv_stride % 64 == 0), at least 256 query rows per item.
bool gar_attn_bf16_v4_try(const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv, int hd, int q_row0,
                          int q_len, int q_total, int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev,
                          const int32_t* kv_start, int kv_prefix, hipStream_t s) {
    if (hd != 64 || q_len < 256 || (int64_t)kv_stride * 128 >= ((int64_t)1 << 31) || (int64_t)q_pad * 128 >= ((int64_t)1 << 31))
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
    int grid = (int)(n_items < cus ? n_items : cus);
#ifdef V4_GRID
    grid = V4_GRID < grid ? V4_GRID : grid;      // timeline builds
#endif
    if (causal) hipLaunchKernelGGL(attn_bf16_v4_kernel<true>, dim3(grid), dim3(256), V4_LDS, s, a);
    else hipLaunchKernelGGL(attn_bf16_v4_kernel<false>, dim3(grid), dim3(256), V4_LDS, s, a);
    return true;
}
