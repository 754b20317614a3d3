// Skinny bf16 GEMM for the decode step: C[M <= 64, N] = epilogue(RMSNorm?(A) W^T), weight-streaming (HBM-bound).
//
//   Block = 16 (32 for SwiGLU) weight rows x all of K; its waves split K; each wave keeps MT = ceil(M/16) MFMA
//   accumulators per weight tile so ONE pass over the weights serves up to 64 sequences (continuous batching of
//   regions amortises the 2.47 GB/token weight stream).
//   Per 64-wide K step lane (row = l&15, g = l>>4) reads W[row][k0+16g .. +16) as two adjacent nontemporal 16-B loads
//   (the four g's of a row cover one 128-B line) and x[16 mt + (l&15)][same k] from L2; MFMA h uses the h-th 8-element
//   half of both. NORM: x*g is the operand and rsqrt(mean(x^2)+eps) scales the accumulator row in the epilogue.
//   The waves' partial accumulators are merged in LDS, row tile mt by wave (mt mod nw).
//   Algorithmic bytes per launch = N*K*2 (weights) + small.
#include <stdlib.h>

#include "gemm_epilogue.h"

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#ifndef SKINNY_WD_NARROW        /* weight-ring depth of the staged path for NT <= 2 (1 = re-arm one step at a time) */
#define SKINNY_WD_NARROW 4
#endif
#ifndef SK_W_AUX                /* cache policy of the weight DMA: 2 = nt. A decode step streams each weight byte once (2.47 GB per
                                   token against 4 MiB of L2 per XCD and a 256 MB Infinity Cache): gate/up -7 %, lm_head -2.5 %,
                                   qkv / o -3 % against the default policy (COLD=1 tools/bench_skinny.py, profiles/r3_decode_nt.txt) */
#define SK_W_AUX 2
#endif
#define SKINNY_WD(NT_) ((NT_) <= 2 ? SKINNY_WD_NARROW : 1)

// STAGED: every wave brings the [16 x 64] bf16 tiles of a K step (NT weight tiles + MT activation tiles, 2 KiB each)
// into a private LDS region with `buffer_load_dwordx4 ... lds` — whole 128-byte row segments, 8 rows per instruction —
// and reads MFMA fragments back with conflict-free ds_read_b128 (chunk ^= (row >> 1) & 7, applied on the DMA source
// side). Per-lane fragment loads from global memory touch 16 lines (half used) per instruction; for M = 64 the four
// activation tiles re-read from L2 that way cost more than the weight stream itself (tools/bench_skinny.py).
// NORM: 0 none; 1 RMSNorm prologue with its gain (p.norm_w): x*g is the operand, sum x^2 accumulated on the VALU; 2 RMSNorm with
// the gain folded into W (p.norm_folded): the operand is x itself and the row sums of squares come off the matrix pipe — one
// extra MFMA pair per row tile and K step, x_tile x_tile^T, whose DIAGONAL is sum_k x[m][k]^2 (HBM-bound kernel: the pipe idles)
// ROWS = 8 (NT == 1 only): a block owns EIGHT weight rows instead of sixteen — the upper half of its MFMA tile is never staged
// nor stored — so that a narrow output (N = 2048: 128 sixteen-row tiles) still gives every CU a block: Llama's o / down at
// M <= 16 stream from 256 CUs instead of 128 (tools/bench_skinny.py, round 4).
// DEEP (ROWS = 8, one row tile of at most EIGHT activation rows, no norm prologue: Llama's o / down at decode batches <= 8): weights AND
// activations ride a ring of SKINNY_DEEP K steps, 1 KiB each (8 rows x 128 B: the upper half of both MFMA tiles is a duplicate of the
// lower one and never stored). The staged path above re-arms the activation tile ONE step ahead; vmcnt retires in order, so waiting
// for it also waits for every weight DMA issued before it and a wave has ~2 KiB in flight however deep its weight ring is —
// `down` (K = 8192: 16 K steps per wave) paid a memory latency every other step: 12.3 us for 33.5 MB at M = 1 (2.7 TB/s).
#ifndef SKINNY_DEEP
#define SKINNY_DEEP 8
#endif
template <int EPI, int NT, int MT, int NORM, bool STAGED, int ROWS = 16, bool DEEP = false>
__global__ __launch_bounds__((NT >= 4 || (NORM == 1 && MT >= 4)) ? 512 : 1024) void skinny_mt_bf16_kernel(const gar_gemm_params p) {
    static_assert(ROWS == 16 || (ROWS == 8 && NT == 1), "half tiles are built for one weight tile per block");
    static_assert(!DEEP || (ROWS == 8 && NT == 1 && MT == 1 && NORM == 0 && STAGED), "the deep ring is built for the half-tile GEMV without a norm prologue");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, nw = blockDim.x >> 6;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float* red = reinterpret_cast<float*>(smem);                  // [nw][NT*MT][64][4]
    float* red_ss = red + nw * NT * MT * 256;                     // [nw][MT][16]
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * ROWS * NT;
    const bf16_t* W = (const bf16_t*)p.W;
    const bf16_t* X = (const bf16_t*)p.A;
    const bf16_t* Gw = (const bf16_t*)p.norm_w;
    // split-K (p.split_k > 1, GAR_EPI_NONE only): blockIdx.y owns one of split_k equal K slices and writes its fp32
    // product to p.partial [split_k][M][N]; gar_splitk_residual_rmsnorm sums the slices (no atomics, no fences: the
    // reduction is the next launch)
    const int nsplit = p.split_k > 1 ? p.split_k : 1;
    const int ksteps = p.K / 64 / nsplit;
    const int kbase = (int)blockIdx.y * ksteps;
    const int per = (ksteps + nw - 1) / nw;
    const int ks0 = kbase + wave * per, ks1 = min(kbase + ksteps, ks0 + per);
    const bf16_t* xp[MT];
    bool xv[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        xv[mt] = mt * 16 + frow < p.M;
        xp[mt] = X + (int64_t)(xv[mt] ? mt * 16 + frow : 0) * p.lda + fq * 16;
    }
    const bf16_t* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + (int64_t)min(n0 + t * 16 + (ROWS == 8 ? (frow & 7) : frow), p.N - 1) * p.ldw + fq * 16;
    f32x4 acc[NT][MT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) acc[t][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float ssq[MT];
    f32x4 ssm[MT];                 // NORM == 2: x_tile x_tile^T accumulators (lane (row, g) holds C[4 g + r][row])
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        ssq[mt] = 0.f;
        ssm[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }

    // ---- STAGED path: DMA tiles -> LDS -> fragments
    // The weight tiles of the next WD - 1 K steps are in flight while a step computes (narrow weight tiles, NT <= 2): a
    // decode GEMM streams COLD weights — 2.47 GB per token against a 256 MB Infinity Cache — so a wave that re-arms one
    // step at a time pays one HBM latency (2-3 us under load) per K step: qkv / o ran 11.6 us inside the decode step
    // against 6.4 us in the micro-benchmark, whose weights stay cache-resident. The activation tiles (L2-resident, the
    // same for every block) keep one slot.
    constexpr int WD = SKINNY_WD(NT);
    constexpr int REG = (WD * NT + MT) * 2048;
    char* wreg = smem + wave * REG;
    char* xreg = wreg + WD * NT * 2048;
    const unsigned rbW = (unsigned)p.ldw * 2u, rbX = (unsigned)p.lda * 2u;
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(
        (void*)W, 0, (int)(((int64_t)(p.N - 1) * p.ldw + p.K) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(
        (void*)X, 0, (int)(((int64_t)(p.M - 1) * p.lda + p.K) * 2), 0x00020000);
    int voffW[2], voffX[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * i + (lane >> 3);
        const int c = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
        voffW[i] = (int)((unsigned)(n0 + row) * rbW) + c;
        voffX[i] = (int)((unsigned)row * rbX) + c;
    }
    auto stage_w = [&](int ks) {
        const unsigned k0b = (unsigned)ks * 128u;
        char* slot = wreg + ((ks - ks0) % WD) * (NT * 2048);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int i = 0; i < (ROWS == 8 ? 1 : 2); ++i)      // ROWS == 8: rows 8..15 of the LDS tile stay zero (cleared once below; their
                                                                 // MFMA output rows are never stored)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_AS(slot + t * 2048 + i * 1024), 16,
                                                         voffW[i] + (int)((unsigned)(t * 16) * rbW + k0b), 0, 0, SK_W_AUX);
    };
    auto stage_x = [&](int ks) {
        const unsigned k0b = (unsigned)ks * 128u;
#ifndef SK_NOX   /* -DSK_NOX: diagnostic build that never stages the activation tiles (wrong results): what the weight
                    stream alone costs (tools/build_variant.sh, profiles/r2_decode_gemm_experiment.txt) */
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_AS(xreg + mt * 2048 + i * 1024), 16,
                                                         voffX[i] + (int)((unsigned)(mt * 16) * rbX + k0b), 0, 0, 0);
#endif
    };
    const int foff0 = frow * 128 + (((2 * fq) ^ ((frow >> 1) & 7)) << 4);
    const int foff1 = frow * 128 + (((2 * fq + 1) ^ ((frow >> 1) & 7)) << 4);
    // issue order: W(ks0), X(ks0), W(ks0+1 .. ks0+WD-1) | per step ks: [wait] X(ks+1), W(ks+WD). vmcnt retires in order, so
    // "all but my newest 2 NT" = everything up to X(ks) has landed — W(ks), issued WD - 1 steps earlier, with it — and the
    // newest weight step stays in flight; the last steps (nothing newer issued behind X) wait for everything.
    auto step_staged = [&](int ks, bool more) {
        const int k0 = ks * 64;
        if (WD > 1 && ks + WD - 1 < ks1 && ks > ks0) {
            if (NT == 1 && ROWS == 8) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else if (NT == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        const char* slot = wreg + ((ks - ks0) % WD) * (NT * 2048);
        bf16x8 w0[NT], w1[NT], xr0[MT], xr1[MT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            w0[t] = *reinterpret_cast<const bf16x8*>(slot + t * 2048 + foff0);
            w1[t] = *reinterpret_cast<const bf16x8*>(slot + t * 2048 + foff1);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            xr0[mt] = *reinterpret_cast<const bf16x8*>(xreg + mt * 2048 + foff0);
            xr1[mt] = *reinterpret_cast<const bf16x8*>(xreg + mt * 2048 + foff1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (more) stage_x(ks + 1);               // re-arm: the fragments of this step are in registers now
        if (ks + WD < ks1) stage_w(ks + WD);
        __builtin_amdgcn_sched_barrier(0);
        float ga[8], gb[8];
        if (NORM == 1) {
            ld8(Gw + fq * 16 + k0, ga);
            ld8(Gw + fq * 16 + k0 + 8, gb);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x8 x0 = xr0[mt], x1 = xr1[mt];
            if (NORM == 2) {
                ssm[mt] = MFMA_16x16x32(x0, x0, ssm[mt]);
                ssm[mt] = MFMA_16x16x32(x1, x1, ssm[mt]);
            }
            if (NORM == 1) {
                const u32x4 ua = __builtin_bit_cast(u32x4, x0), ub = __builtin_bit_cast(u32x4, x1);
                u32x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float a0 = unpk_lo(ua[e]), a1 = unpk_hi(ua[e]);
                    const float b0 = unpk_lo(ub[e]), b1 = unpk_hi(ub[e]);
                    ssq[mt] += a0 * a0 + a1 * a1 + b0 * b0 + b1 * b1;
                    pa[e] = pack_bf2(a0 * ga[2 * e], a1 * ga[2 * e + 1]);
                    pb[e] = pack_bf2(b0 * gb[2 * e], b1 * gb[2 * e + 1]);
                }
                x0 = __builtin_bit_cast(bf16x8, pa);
                x1 = __builtin_bit_cast(bf16x8, pb);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][mt] = MFMA_16x16x32(w0[t], x0, acc[t][mt]);
                acc[t][mt] = MFMA_16x16x32(w1[t], x1, acc[t][mt]);
            }
        }
    };

    auto step = [&](int ks) {
        const int k0 = ks * 64;
        bf16x8 w0[NT], w1[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            w0[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + k0));
            w1[t] = __builtin_nontemporal_load(reinterpret_cast<const bf16x8*>(wp[t] + k0 + 8));
        }
        float ga[8], gb[8];
        if (NORM == 1) {
            ld8(Gw + fq * 16 + k0, ga);
            ld8(Gw + fq * 16 + k0 + 8, gb);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            bf16x8 x0, x1;
            if (NORM == 1) {
                float xa[8], xb[8];
                ld8(xp[mt] + k0, xa);
                ld8(xp[mt] + k0 + 8, xb);
                u32x4 pa, pb;
#pragma unroll
                for (int e = 0; e < 8; e += 2) {
                    ssq[mt] += xa[e] * xa[e] + xa[e + 1] * xa[e + 1] + xb[e] * xb[e] + xb[e + 1] * xb[e + 1];
                    pa[e >> 1] = pack_bf2(xa[e] * ga[e], xa[e + 1] * ga[e + 1]);
                    pb[e >> 1] = pack_bf2(xb[e] * gb[e], xb[e + 1] * gb[e + 1]);
                }
                x0 = __builtin_bit_cast(bf16x8, pa);
                x1 = __builtin_bit_cast(bf16x8, pb);
            } else {
                x0 = *reinterpret_cast<const bf16x8*>(xp[mt] + k0);
                x1 = *reinterpret_cast<const bf16x8*>(xp[mt] + k0 + 8);
                if (NORM == 2) {
                    ssm[mt] = MFMA_16x16x32(x0, x0, ssm[mt]);
                    ssm[mt] = MFMA_16x16x32(x1, x1, ssm[mt]);
                }
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t][mt] = MFMA_16x16x32(w0[t], x0, acc[t][mt]);
                acc[t][mt] = MFMA_16x16x32(w1[t], x1, acc[t][mt]);
            }
        }
    };
    int ks = ks0;
    if constexpr (DEEP) {
        constexpr int DW = SKINNY_DEEP;
        char* ring = smem + wave * (DW * 2048);
        const int drow = lane >> 3, dch = ((lane & 7) ^ ((drow >> 1) & 7)) << 4;
        const int dvW = (int)((unsigned)(n0 + drow) * rbW) + dch, dvX = (int)((unsigned)drow * rbX) + dch;
        auto issue = [&](int k_) {
            char* slot = ring + ((k_ - ks0) % DW) * 2048;
            const unsigned k0b = (unsigned)k_ * 128u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, LDS_AS(slot), 16, dvW + (int)k0b, 0, 0, SK_W_AUX);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsX, LDS_AS(slot + 1024), 16, dvX + (int)k0b, 0, 0, 0);
        };
        // fragment rows 8..15 read rows 0..7 again (defined data; their outputs are never stored)
        const int fr = frow & 7;
        const int d0 = fr * 128 + (((2 * fq) ^ ((fr >> 1) & 7)) << 4), d1 = fr * 128 + (((2 * fq + 1) ^ ((fr >> 1) & 7)) << 4);
#pragma unroll
        for (int d = 0; d < DW - 1; ++d)
            if (ks0 + d < ks1) issue(ks0 + d);
        for (; ks < ks1; ++ks) {
            if (ks + DW - 1 < ks1) issue(ks + DW - 1);
            // everything up to step ks has landed; the steps issued behind it (two DMA instructions each) stay in flight
            const int rem = min(DW - 1, ks1 - 1 - ks);
            if (__builtin_expect(rem == DW - 1, 1)) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * (DW - 1)) : "memory");
            else if (rem >= 6) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else if (rem == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (rem == 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (rem == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else if (rem == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (rem == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const char* slot = ring + ((ks - ks0) % DW) * 2048;
            const bf16x8 w0 = *reinterpret_cast<const bf16x8*>(slot + d0), w1 = *reinterpret_cast<const bf16x8*>(slot + d1);
            const bf16x8 x0 = *reinterpret_cast<const bf16x8*>(slot + 1024 + d0), x1 = *reinterpret_cast<const bf16x8*>(slot + 1024 + d1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = MFMA_16x16x32(w0, x0, acc[0][0]);
            acc[0][0] = MFMA_16x16x32(w1, x1, acc[0][0]);
        }
        __syncthreads();                          // the merge buffer below aliases the rings
    } else if (STAGED) {
        if (ROWS == 8) {
            // rows 8..15 of every weight slot are never staged: zero them once so that the MFMAs whose output rows are discarded
            // consume defined operands (no NaN / Inf bit patterns out of whatever the LDS held). One 1-KiB store per
            // slot and wave; LDS operations of a wave execute in order, so the fragment reads below see it.
#pragma unroll
            for (int d = 0; d < WD; ++d)
                *reinterpret_cast<u32x4*>(wreg + d * (NT * 2048) + 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
        }
        if (ks < ks1) {
            stage_w(ks);
            stage_x(ks);
#pragma unroll
            for (int d = 1; d < WD; ++d)
                if (ks + d < ks1) stage_w(ks + d);
        }
        for (; ks < ks1; ++ks) step_staged(ks, ks + 1 < ks1);
        __syncthreads();                          // the merge buffer below aliases the staging regions
    } else {
        for (; ks + 2 <= ks1; ks += 2) {
            step(ks);
            step(ks + 1);
        }
        for (; ks < ks1; ++ks) step(ks);
    }

#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
            *reinterpret_cast<f32x4*>(red + (((wave * NT + t) * MT + mt) * 64 + lane) * 4) = acc[t][mt];
    if (NORM == 1) {
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            float s = ssq[mt];
            s += __shfl_xor(s, 16, 64);
            s += __shfl_xor(s, 32, 64);
            if (lane < 16) red_ss[(wave * MT + mt) * 16 + lane] = s;
        }
    }
    if (NORM == 2) {          // the diagonal: row i = 4 g + r of column i sits in lane (row = i, g = i >> 2), register i & 3
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const f32x4 d = ssm[mt];
            const int r = frow & 3;
            const float s = r == 0 ? d[0] : (r == 1 ? d[1] : (r == 2 ? d[2] : d[3]));
            if (fq == (frow >> 2)) red_ss[(wave * MT + mt) * 16 + frow] = s;
        }
    }
    __syncthreads();
    // row tile mt is finished by wave (mt mod nw)
    for (int mt = wave; mt < MT; mt += nw) {
        const int m = mt * 16 + frow;
        float v[NT][4];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[t][r] = 0.f;
            for (int w = 0; w < nw; ++w) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(red + (((w * NT + t) * MT + mt) * 64 + lane) * 4);
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t][r] += a[r];
            }
        }
        if (NORM) {
            float tot = 0.f;
            for (int w = 0; w < nw; ++w) tot += red_ss[(w * MT + mt) * 16 + frow];
            const float rstd = rsqrtf(tot / (float)p.K + p.norm_eps);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) v[t][r] *= rstd;
        }
        if (m >= p.M) continue;
        if (ROWS == 8 && fq >= 2) continue;            // weight rows 8..15 of a half tile belong to the next block
        if (EPI == GAR_EPI_NONE && nsplit > 1) {
            float* part = reinterpret_cast<float*>(p.partial) + ((int64_t)blockIdx.y * p.M + m) * p.N;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + t * 16 + fq * 4;
                if (n < p.N) *reinterpret_cast<f32x4*>(part + n) = f32x4{v[t][0], v[t][1], v[t][2], v[t][3]};
            }
            continue;
        }
        if (EPI == GAR_EPI_SWIGLU) {          // weight tiles come in (gate16, up16) pairs
#pragma unroll
            for (int q = 0; q < NT / 2; ++q) {
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = silu_fast(v[2 * q][r]) * v[2 * q + 1][r];
                if (n0 + q * 32 < p.N) epilogue_store<bf16_t, EPI>(p, m, (n0 >> 1) + q * 16 + fq * 4, o);
            }
        } else {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int n = n0 + t * 16 + fq * 4;
                if (n < p.N) epilogue_store<bf16_t, EPI>(p, m, n, v[t]);
            }
        }
    }
}

template <int EPI, int MT, int NT, int ROWS = 16, bool DEEP = false>
static void launch_skinny(const gar_gemm_params& p, hipStream_t s) {
    const int nb = (p.N + ROWS * NT - 1) / (ROWS * NT);
    const int nsplit = (EPI == GAR_EPI_NONE && p.split_k > 1) ? p.split_k : 1;
    const int ksteps = p.K / 64 / nsplit;
    // buffer descriptors address W / x with 32-bit byte offsets (larger operands take the per-lane fragment loads)
    const bool staged = ((int64_t)(p.N - 1) * p.ldw + p.K) * 2 < ((int64_t)1 << 31) &&
                        ((int64_t)(p.M - 1) * p.lda + p.K) * 2 < ((int64_t)1 << 31);
    constexpr int MAXLDS = 139264;
    static gar_once_per_device attr_once;
    attr_once.run([&] {
        if constexpr (NT <= 2) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, NT, MT, 1, true, ROWS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, NT, MT, 1, false, ROWS>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
        }
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, NT, MT, 0, true, ROWS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, NT, MT, 0, false, ROWS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, NT, MT, 2, true, ROWS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, NT, MT, 2, false, ROWS>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
    });
    // enough waves to cover HBM latency (>= ~2048 chip-wide), >= 2 K steps per wave, LDS (merge buffer, and the
    // per-wave staging regions it aliases) <= 136 KiB
    auto lds_for = [&](int w) {
        const int merge = w * NT * MT * 1024 + w * MT * 64;
        const int stg = staged ? w * (SKINNY_WD(NT) * NT + MT) * 2048 : 0;
        return merge > stg ? merge : stg;
    };
    int nw = 4;
    // 4-tile blocks and the fused-norm 4-row-tile blocks are built for <= 512 threads (256 VGPRs)
    const int max_nw = (NT >= 4 || (p.norm_w && MT >= 4)) ? 8 : 16;
    while (nw < max_nw && nb * nsplit * nw < 2048 && ksteps / (nw * 2) >= 2 && lds_for(nw * 2) <= MAXLDS) nw *= 2;
    const int lds = lds_for(nw);
#define LAUNCH_SK(NORM_, ST_) \
    hipLaunchKernelGGL((skinny_mt_bf16_kernel<EPI, NT, MT, NORM_, ST_, ROWS>), dim3(nb, nsplit), dim3(nw * 64), lds, s, p)
    if constexpr (NT <= 2) {
        if (p.norm_w) {
            if (staged) LAUNCH_SK(1, true); else LAUNCH_SK(1, false);
            return;
        }
    }
    if (p.norm_folded) {
        if (staged) LAUNCH_SK(2, true); else LAUNCH_SK(2, false);
        return;
    }
    if (staged) LAUNCH_SK(0, true); else LAUNCH_SK(0, false);
#undef LAUNCH_SK
}

// the deep-ring half-tile GEMV (DEEP): M <= 8 rows, narrow output, no norm prologue. false = not applicable.
template <int EPI>
static bool launch_skinny_deep(const gar_gemm_params& p, hipStream_t s) {
    const bool staged = ((int64_t)(p.N - 1) * p.ldw + p.K) * 2 < ((int64_t)1 << 31) && ((int64_t)(p.M - 1) * p.lda + p.K) * 2 < ((int64_t)1 << 31);
    if (!staged || p.M > 8 || p.norm_w || p.norm_folded || p.split_k > 1) return false;
    const int nb = (p.N + 7) / 8, ksteps = p.K / 64;
    constexpr int MAXLDS = 139264;
    static gar_once_per_device attr_once;
    attr_once.run([&] {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&skinny_mt_bf16_kernel<EPI, 1, 1, 0, true, 8, true>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, MAXLDS);
    });
    int nw = 4;
    while (nw < 8 && nb * nw < 2048 && ksteps / (nw * 2) >= 2) nw *= 2;
    const int lds = nw * SKINNY_DEEP * 2048;        // >= the merge buffer (nw x 1088 B)
    hipLaunchKernelGGL((skinny_mt_bf16_kernel<EPI, 1, 1, 0, true, 8, true>), dim3(nb, 1), dim3(nw * 64), lds, s, p);
    return true;
}

// NT = weight tiles (16 rows) per block. With several row tiles (M > 16) every block re-reads the activations from
// L2, MT x the bytes of its weight tile: wide outputs (gate/up, lm_head) use 4 weight tiles per block so one activation
// fragment serves 4 of them; narrow outputs keep 1 (2 for SwiGLU pairs) for the sake of block count.
template <int EPI>
static void launch_skinny_m(const gar_gemm_params& p, hipStream_t s) {
    constexpr int NT0 = (EPI == GAR_EPI_SWIGLU) ? 2 : 1;
    constexpr int wide = 8192;      // output width from which M > 16 uses 4 weight tiles per block (tools/bench_skinny.py)
#ifndef SK_HALF_MAX_N       /* widest output that takes 8-row half tiles at M <= 16 (0 = never) */
#define SK_HALF_MAX_N 2048
#endif
    if (p.M <= 16) {
        // a narrow output: 8-row half tiles so that N / 8 blocks (256 for N = 2048) cover the chip instead of N / 16
        if (NT0 == 1 && p.N <= SK_HALF_MAX_N && p.N % 8 == 0 && p.split_k <= 1) {
            if constexpr (NT0 == 1) {
                if (!launch_skinny_deep<EPI>(p, s)) launch_skinny<EPI, 1, 1, 8>(p, s);
            }
        } else {
            launch_skinny<EPI, 1, NT0>(p, s);
        }
    }
    else if (p.M <= 32) {
        if (p.N >= wide && !p.norm_w) launch_skinny<EPI, 2, 4>(p, s); else launch_skinny<EPI, 2, NT0>(p, s);
    } else {
        // more 16-row weight tiles than CUs (a block's staging regions fill a CU's LDS: one block per CU): two tiles per
        // block keep the grid to one round and halve the activation re-reads — Llama-3.1-8B's qkv (N = 6144, 384 tiles)
        // 24.6 -> 18.2 us with cold weights; at N = 4096 (256 tiles) and below one tile per block stays faster
        // (split-K slices count as blocks: `down` at N = 4096 x 2 slices, or N = 2048 x 4, also fills the chip with two-tile
        // blocks — four row tiles re-read per ONE weight tile is 4 bytes of L2 traffic per weight byte, round 4)
        const int nsp = (EPI == GAR_EPI_NONE && p.split_k > 1) ? p.split_k : 1;
        if (p.N >= wide && !p.norm_w) launch_skinny<EPI, 4, 4>(p, s);
        else if ((p.N > 16 * 256 || (p.N / 32) * nsp >= 256) && !p.norm_w) launch_skinny<EPI, 4, 2>(p, s);
        else launch_skinny<EPI, 4, NT0>(p, s);
    }
}

// bf16, M <= 64, the epilogues the decode step uses. Returns false otherwise.
bool gar_skinny_bf16_try(const gar_gemm_params& p, hipStream_t s) {
    if (p.M > 64) return false;
    switch (p.epilogue) {
        case GAR_EPI_NONE: launch_skinny_m<GAR_EPI_NONE>(p, s); return true;
        case GAR_EPI_RES: launch_skinny_m<GAR_EPI_RES>(p, s); return true;
        case GAR_EPI_SWIGLU: launch_skinny_m<GAR_EPI_SWIGLU>(p, s); return true;
        case GAR_EPI_BIAS: launch_skinny_m<GAR_EPI_BIAS>(p, s); return true;
        default: return false;
    }
}
