// LayerNorm / RMSNorm — one wave per row, 8 elements (16 B bf16 / 32 B f32) per lane per step, fp32 statistics.
// HBM-bound: algorithmic bytes = 2 * M * D * sizeof(T).
#include <stdlib.h>

#include "common.h"

// No implicit FMA contraction in this file: hipcc's default (-ffp-contract=fast) fused the rows a wave handles
// differently — a row's LayerNorm then depended (by one bf16 ulp, in a handful of elements per million) on whether it was
// the first or the second row of its wave, i.e. on the position of its sample in the batch (found with two identical
// samples in one batch, tools/debug_vit_bisect.py). Every fused multiply-add below is an explicit __fmaf_rn, written
// once (ln_out / rms_out / the accumulation helpers) and used by both kernels.
#pragma clang fp contract(off)

__device__ __forceinline__ float ln_out(float v, float mean, float rstd, float w, float b) {
    return __fmaf_rn((v - mean) * rstd, w, b);
}
template <typename T>
__device__ __forceinline__ float rms_out(float v, float rstd, float w) {
    float n = v * rstd;
    // HF LlamaRMSNorm: weight * hidden.to(input_dtype) -> the normalised value is rounded to the storage dtype before the
    // weight multiply
    if (sizeof(T) == 2) n = bf2f(f2bf(n));
    return w * n;
}

// MAXC = register-cached chunks of 512 elements (rows up to 512*MAXC stay in registers: one HBM read); instantiated
// for 2 / 4 / 8 so short rows (ViT D=1024) keep the VGPR count — and with it the occupancy that hides HBM latency — low

template <typename T, bool RMS, int NORM_MAXC>
__global__ __launch_bounds__(256) void norm_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ w,
                                                   const T* __restrict__ b, int M, int D, int64_t ldx, int64_t ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + (int64_t)row * ldx;
    T* yr = y + (int64_t)row * ldy;
    const int nchunk = (D + 511) / 512;
    float v[NORM_MAXC][8];
    float s = 0.f;
    const bool cached = nchunk <= NORM_MAXC;
    if (cached) {
#pragma unroll
        for (int c = 0; c < NORM_MAXC; ++c) {
            const int i = c * 512 + lane * 8;
            if (c < nchunk && i < D) {
                ld8(xr + i, v[c]);
#pragma unroll
                for (int e = 0; e < 8; ++e) s = RMS ? __fmaf_rn(v[c][e], v[c][e], s) : s + v[c][e];
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[c][e] = 0.f;
            }
        }
    } else {
        for (int i = lane * 8; i < D; i += 512) {
            float t[8];
            ld8(xr + i, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) s = RMS ? __fmaf_rn(t[e], t[e], s) : s + t[e];
        }
    }
    s = wave_sum(s);
    float mean = 0.f, rstd;
    if (RMS) {
        rstd = rsqrtf(s / (float)D + eps);
    } else {
        mean = s / (float)D;
        float q = 0.f;
        if (cached) {
#pragma unroll
            for (int c = 0; c < NORM_MAXC; ++c) {
                const int i = c * 512 + lane * 8;
                if (c < nchunk && i < D) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q = __fmaf_rn(d, d, q); }
                }
            }
        } else {
            for (int i = lane * 8; i < D; i += 512) {
                float t[8];
                ld8(xr + i, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = t[e] - mean; q = __fmaf_rn(d, d, q); }
            }
        }
        q = wave_sum(q);
        rstd = rsqrtf(q / (float)D + eps);
    }
    auto emit = [&](int i, float (&t)[8]) {
        float ww[8], o[8];
        ld8(w + i, ww);
        if (RMS) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rms_out<T>(t[e], rstd, ww[e]);
        } else {
            float bb[8];
            ld8(b + i, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = ln_out(t[e], mean, rstd, ww[e], bb[e]);
        }
        st8(yr + i, o);
    };
    if (cached) {
#pragma unroll
        for (int c = 0; c < NORM_MAXC; ++c) {
            const int i = c * 512 + lane * 8;
            if (c < nchunk && i < D) emit(i, v[c]);
        }
    } else {
        for (int i = lane * 8; i < D; i += 512) {
            float t[8];
            ld8(xr + i, t);
            emit(i, t);
        }
    }
}

// R rows per wave, cached RAW (bf16, 4 VGPRs per 8 elements) and unpacked where they are used: a CU of this part holds 16
// waves (profiles/r2_occupancy_probe.txt), so the bytes in flight that cover HBM latency have to come from inside the
// wave — 8 x 16 B per lane here (R = 4 rows of D <= 1024, R = 2 of D <= 2048) against 4 with fp32-cached rows. Same
// arithmetic, in the same order, as norm_kernel: a row's result does not depend on its place in the wave.
__device__ __forceinline__ void unpack8(const uint4& t, float (&v)[8]) {
    v[0] = unpk_lo(t.x); v[1] = unpk_hi(t.x);
    v[2] = unpk_lo(t.y); v[3] = unpk_hi(t.y);
    v[4] = unpk_lo(t.z); v[5] = unpk_hi(t.z);
    v[6] = unpk_lo(t.w); v[7] = unpk_hi(t.w);
}

template <bool RMS, int NC, int R>
__global__ __launch_bounds__(256) void normr_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                    const bf16_t* __restrict__ w, const bf16_t* __restrict__ b, int M,
                                                    int D, int64_t ldx, int64_t ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R;
    if (row0 >= M) return;
    uint4 raw[R][NC], wr[NC], br[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 512 + lane * 8;
        const bool in = i < D;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int rr = min(row0 + r, M - 1);
            raw[r][c] = in ? *reinterpret_cast<const uint4*>(x + (int64_t)rr * ldx + i) : make_uint4(0u, 0u, 0u, 0u);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 512 + lane * 8;
        wr[c] = i < D ? *reinterpret_cast<const uint4*>(w + i) : make_uint4(0u, 0u, 0u, 0u);
        if (!RMS) br[c] = i < D ? *reinterpret_cast<const uint4*>(b + i) : make_uint4(0u, 0u, 0u, 0u);
    }
    float s[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        s[r] = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            float v[8];
            unpack8(raw[r][c], v);
#pragma unroll
            for (int e = 0; e < 8; ++e) s[r] = RMS ? __fmaf_rn(v[e], v[e], s[r]) : s[r] + v[e];
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int r = 0; r < R; ++r) s[r] += __shfl_xor(s[r], o, 64);
    float mean[R], rstd[R];
    if (RMS) {
#pragma unroll
        for (int r = 0; r < R; ++r) { mean[r] = 0.f; rstd[r] = rsqrtf(s[r] / (float)D + eps); }
    } else {
        float q[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            mean[r] = s[r] / (float)D;
            q[r] = 0.f;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c * 512 + lane * 8 < D) {
                    float v[8];
                    unpack8(raw[r][c], v);
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean[r]; q[r] = __fmaf_rn(d, d, q[r]); }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < R; ++r) q[r] += __shfl_xor(q[r], o, 64);
#pragma unroll
        for (int r = 0; r < R; ++r) rstd[r] = rsqrtf(q[r] / (float)D + eps);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        if (row0 + r >= M) break;
        bf16_t* yr = y + (int64_t)(row0 + r) * ldy;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = c * 512 + lane * 8;
            if (i >= D) continue;
            float v[8], ww[8], bb[8], o[8];
            unpack8(raw[r][c], v);
            unpack8(wr[c], ww);
            if (!RMS) unpack8(br[c], bb);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = RMS ? rms_out<bf16_t>(v[e], rstd[r], ww[e]) : ln_out(v[e], mean[r], rstd[r], ww[e], bb[e]);
            st8(yr + i, o);
        }
    }
}

template <bool RMS>
static int launch_norm(int dtype, const void* x, void* y, const void* w, const void* b, int M, int D, int64_t ldx,
                       int64_t ldy, float eps, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "norm: bad dtype");
    GAR_CHECK_ARG(x && y && w && (RMS || b), "norm: null pointer");
    GAR_CHECK_ARG(M > 0 && D > 0 && D % 8 == 0, "norm: D=%d must be a multiple of 8", D);
    if (ldx <= 0) ldx = D;
    if (ldy <= 0) ldy = D;
    GAR_CHECK_ARG(ldx % 8 == 0 && ldy % 8 == 0, "norm: row strides must be multiples of 8 elements");
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_NORM(TT, C_)                                                                                    \
    hipLaunchKernelGGL((norm_kernel<TT, RMS, C_>), grid, block, 0, s, (const TT*)x, (TT*)y, (const TT*)w, (const TT*)b, M, \
                       D, ldx, ldy, eps)
    // bf16 rows that fit 2 register chunks and enough rows to fill the chip: two rows per wave (ViT LN 3.8 -> 4.5 TB/s)
    // bf16 rows that fit the register cache and enough rows to fill the chip (ViT LN 5.2 -> 5.9 TB/s, RMSNorm 5.6 -> 5.7)
    if (dtype == GAR_BF16 && D <= 2048 && M >= 4096) {
#define LAUNCH_NORMR(C_, R_)                                                                                     \
    hipLaunchKernelGGL((normr_kernel<RMS, C_, R_>), dim3((M + 4 * R_ - 1) / (4 * R_)), block, 0, s, (const bf16_t*)x, \
                       (bf16_t*)y, (const bf16_t*)w, (const bf16_t*)b, M, D, ldx, ldy, eps)
        if (D <= 1024) LAUNCH_NORMR(2, 4); else LAUNCH_NORMR(4, 2);
#undef LAUNCH_NORMR
        GAR_CHECK_LAUNCH();
        return GAR_OK;
    }
    if (dtype == GAR_BF16) {
        if (D <= 1024) LAUNCH_NORM(bf16_t, 2); else if (D <= 2048) LAUNCH_NORM(bf16_t, 4); else LAUNCH_NORM(bf16_t, 8);
    } else {
        if (D <= 1024) LAUNCH_NORM(float, 2); else if (D <= 2048) LAUNCH_NORM(float, 4); else LAUNCH_NORM(float, 8);
    }
#undef LAUNCH_NORM
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// Reduction of a split-K decode GEMM (gemm_skinny.hip, p.split_k > 1) + residual add + the RMSNorm that follows it in a
// Llama layer, one wave per row (M <= 64). The residual stream is rounded to bf16 exactly where GAR_EPI_RES rounds it, and
// the norm reads the ROUNDED row with norm_kernel's arithmetic (same chunk / lane / element order), so this launch gives
// what `gemm(EPI_RES)` + `rmsnorm` give up to the fp32 summation order of the K slices.
template <int NC, int SMAX>
__global__ __launch_bounds__(256) void splitk_res_rms_kernel(const float* __restrict__ part, int S, bf16_t* __restrict__ h,
                                                             const bf16_t* __restrict__ w, bf16_t* __restrict__ y, int M,
                                                             int D, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    bf16_t* hr = h + (int64_t)row * D;
    // every load of the row (S slices x NC chunks + the residual + the weight) is issued before the first add: the launch
    // is 64 waves on an otherwise idle chip, i.e. pure latency
    float4 pa[NC][SMAX][2];
    uint4 hraw[NC], wraw[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 512 + lane * 8;
        const bool in = i < D;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            const bool on = in && s < S;
            const float4* src = reinterpret_cast<const float4*>(part + ((int64_t)(on ? s : 0) * M + row) * D + (in ? i : 0));
            pa[c][s][0] = on ? src[0] : make_float4(0.f, 0.f, 0.f, 0.f);
            pa[c][s][1] = on ? src[1] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        hraw[c] = in ? *reinterpret_cast<const uint4*>(hr + i) : make_uint4(0u, 0u, 0u, 0u);
        wraw[c] = (in && y) ? *reinterpret_cast<const uint4*>(w + i) : make_uint4(0u, 0u, 0u, 0u);
    }
    float v[NC][8];
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 512 + lane * 8;
        float acc[8], r[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            if (s < S) {                                   // slices in order; a skipped slice adds nothing (not even +0)
                const float t[8] = {pa[c][s][0].x, pa[c][s][0].y, pa[c][s][0].z, pa[c][s][0].w,
                                    pa[c][s][1].x, pa[c][s][1].y, pa[c][s][1].z, pa[c][s][1].w};
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = acc[e] + t[e];
            }
        }
        unpack8(hraw[c], r);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[c][e] = i < D ? bf2f(f2bf(r[e] + acc[e])) : 0.f;
        if (i < D) st8(hr + i, v[c]);
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = __fmaf_rn(v[c][e], v[c][e], ss);
    }
    if (!y) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float rstd = rsqrtf(ss / (float)D + eps);
    bf16_t* yr = y + (int64_t)row * D;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 512 + lane * 8;
        if (i >= D) continue;
        float ww[8], o[8];
        unpack8(wraw[c], ww);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = rms_out<bf16_t>(v[c][e], rstd, ww[e]);
        st8(yr + i, o);
    }
}

extern "C" int gar_splitk_residual_rmsnorm(int dtype, const float* partial, int split_k, void* h, const void* w, void* y,
                                           int M, int D, float eps, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_BF16, "splitk_residual_rmsnorm: bf16 only");
    GAR_CHECK_ARG(partial && h && (w || !y), "splitk_residual_rmsnorm: null pointer");
    GAR_CHECK_ARG(split_k >= 1 && split_k <= (D <= 2048 ? 8 : 4) && M > 0 && M <= 64 && D > 0 && D % 8 == 0 && D <= 4096,
                  "splitk_residual_rmsnorm: bad shape (split_k %d, M %d, D %d)", split_k, M, D);
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_SKR(C_)                                                                                              \
    do {                                                                                                            \
        if (split_k <= 4)                                                                                           \
            hipLaunchKernelGGL((splitk_res_rms_kernel<C_, 4>), grid, block, 0, s, partial, split_k, (bf16_t*)h,     \
                               (const bf16_t*)w, (bf16_t*)y, M, D, eps);                                            \
        else                                                                                                        \
            hipLaunchKernelGGL((splitk_res_rms_kernel<C_, 8>), grid, block, 0, s, partial, split_k, (bf16_t*)h,     \
                               (const bf16_t*)w, (bf16_t*)y, M, D, eps);                                            \
    } while (0)
    if (D <= 1024) LAUNCH_SKR(2);
    else if (D <= 2048) LAUNCH_SKR(4);
    else hipLaunchKernelGGL((splitk_res_rms_kernel<8, 4>), grid, block, 0, s, partial, split_k, (bf16_t*)h,
                            (const bf16_t*)w, (bf16_t*)y, M, D, eps);
#undef LAUNCH_SKR
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_layernorm(int dtype, const void* x, void* y, const void* w, const void* b, int M, int D, int64_t ldx,
                             int64_t ldy, float eps, gar_stream_t stream) {
    return launch_norm<false>(dtype, x, y, w, b, M, D, ldx, ldy, eps, stream);
}

extern "C" int gar_rmsnorm(int dtype, const void* x, void* y, const void* w, int M, int D, int64_t ldx, int64_t ldy,
                           float eps, gar_stream_t stream) {
    return launch_norm<true>(dtype, x, y, w, nullptr, M, D, ldx, ldy, eps, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Statistics half of a norm folded into the GEMM pair around it (gar_gemm_params.row_scale / row_stats, ABI v8).
// row_rstd_kernel: one wave per row straight from x (two-pass variance like norm_kernel) — the first norm of a chain.
// row_stats_finalize_kernel: 16 lanes per row over the producer GEMM's (sum, sum of squares) partials, summed in a fixed
// order (deterministic, independent of how the rows were tiled).
// ---------------------------------------------------------------------------------------------------------------
template <typename T, bool RMS>
__global__ __launch_bounds__(256) void row_rstd_kernel(const T* __restrict__ x, int M, int D, int64_t ldx, float eps,
                                                       float* __restrict__ rstd) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const T* xr = x + (int64_t)row * ldx;
    float s = 0.f;
    for (int i = lane * 8; i < D; i += 512) {
        float t[8];
        ld8(xr + i, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) s = RMS ? __fmaf_rn(t[e], t[e], s) : s + t[e];
    }
    s = wave_sum(s);
    float r;
    if (RMS) {
        r = rsqrtf(s / (float)D + eps);
    } else {
        const float mean = s / (float)D;
        float q = 0.f;
        for (int i = lane * 8; i < D; i += 512) {
            float t[8];
            ld8(xr + i, t);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = t[e] - mean; q = __fmaf_rn(d, d, q); }
        }
        q = wave_sum(q);
        r = rsqrtf(q / (float)D + eps);
    }
    if (lane == 0) rstd[row] = r;
}

// 16 lanes per row: lane j sums strips j, j + 16, .. (each load instruction reads 128 contiguous bytes per row — one thread
// per row walked 16 lines per instruction and ran at 0.75 TB/s), then a fixed butterfly over the 16 lanes
__global__ __launch_bounds__(256) void row_stats_finalize_kernel(const float* __restrict__ stats, int M, int strips, int D,
                                                                 float eps, int rms, float* __restrict__ rstd) {
    const int j = threadIdx.x & 15;
    const int row = blockIdx.x * 16 + (threadIdx.x >> 4);
    const bool live = row < M;
    const float2* p = reinterpret_cast<const float2*>(stats) + (int64_t)(live ? row : 0) * strips;
    float s1 = 0.f, s2 = 0.f;
    for (int k = j; k < strips; k += 16) {
        const float2 v = p[k];
        s1 += v.x;
        s2 += v.y;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
        s1 += __shfl_xor(s1, o, 64);
        s2 += __shfl_xor(s2, o, 64);
    }
    const float inv = 1.0f / (float)D;
    float var = s2 * inv;
    if (!rms) {
        const float mean = s1 * inv;
        var = fmaxf(var - mean * mean, 0.f);
    }
    if (live && j == 0) rstd[row] = rsqrtf(var + eps);
}

extern "C" int gar_row_rstd(int dtype, const void* x, int M, int D, int64_t ldx, float eps, int rms, float* rstd,
                            gar_stream_t stream) {
    GAR_CHECK_ARG(x && rstd && M > 0 && D > 0 && D % 8 == 0 && ldx >= D, "row_rstd: bad args");
    GAR_CHECK_ARG(dtype == GAR_BF16 || dtype == GAR_F32, "row_rstd: bad dtype");
    dim3 grid((M + 3) / 4), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_RR(TT, R_) hipLaunchKernelGGL((row_rstd_kernel<TT, R_>), grid, block, 0, s, (const TT*)x, M, D, ldx, eps, rstd)
    if (dtype == GAR_BF16) { if (rms) LAUNCH_RR(bf16_t, true); else LAUNCH_RR(bf16_t, false); }
    else { if (rms) LAUNCH_RR(float, true); else LAUNCH_RR(float, false); }
#undef LAUNCH_RR
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int gar_row_stats_finalize(const float* row_stats, int M, int strips, int D, float eps, int rms, float* rstd,
                                      gar_stream_t stream) {
    GAR_CHECK_ARG(row_stats && rstd && M > 0 && strips > 0 && D > 0, "row_stats_finalize: bad args");
    hipLaunchKernelGGL(row_stats_finalize_kernel, dim3((M + 15) / 16), dim3(256), 0, (hipStream_t)stream, row_stats, M,
                       strips, D, eps, rms, rstd);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
