// LayerNorm / RMSNorm — one wave per row, 8 elements (16 B bf16 / 32 B f32) per lane per step, fp32 statistics.
// HBM-bound: algorithmic bytes = 2 * M * D * sizeof(T).
#include <stdlib.h>

#include "common.h"

// No implicit FMA contraction in this file: hipcc's default (-ffp-contract=fast) fused the two rows of norm2_kernel
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

// Two rows per wave, register-cached (D <= 512*NC): both rows' loads and the weight / bias vectors (shared by the two
// rows) are issued up front, so a wave has twice the bytes in flight and the gamma/beta fetch is off the critical
// path between the statistics and the stores.
template <typename T, bool RMS, int NC>
__global__ __launch_bounds__(256) void norm2_kernel(const T* __restrict__ x, T* __restrict__ y, const T* __restrict__ w,
                                                    const T* __restrict__ b, int M, int D, int64_t ldx, int64_t ldy,
                                                    float eps) {
    const int lane = threadIdx.x & 63;
    const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 2;
    if (row0 >= M) return;
    const bool two = row0 + 1 < M;
    const T* xr[2] = {x + (int64_t)row0 * ldx, x + (int64_t)(two ? row0 + 1 : row0) * ldx};
    float v[2][NC][8], ww[NC][8], bb[NC][8];
    float s[2] = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int i = c * 512 + lane * 8;
        const bool in = i < D;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (in) ld8(xr[r] + i, v[r][c]);
            else {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[r][c][e] = 0.f;
            }
        }
        if (in) {
            ld8(w + i, ww[c]);
            if (!RMS) ld8(b + i, bb[c]);
        }
    }
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) s[r] = RMS ? __fmaf_rn(v[r][c][e], v[r][c][e], s[r]) : s[r] + v[r][c][e];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s[0] += __shfl_xor(s[0], o, 64);
        s[1] += __shfl_xor(s[1], o, 64);
    }
    float mean[2] = {0.f, 0.f}, rstd[2];
    if (RMS) {
        rstd[0] = rsqrtf(s[0] / (float)D + eps);
        rstd[1] = rsqrtf(s[1] / (float)D + eps);
    } else {
        float q[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            mean[r] = s[r] / (float)D;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                if (c * 512 + lane * 8 < D) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) { const float d = v[r][c][e] - mean[r]; q[r] = __fmaf_rn(d, d, q[r]); }
                }
            }
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            q[0] += __shfl_xor(q[0], o, 64);
            q[1] += __shfl_xor(q[1], o, 64);
        }
        rstd[0] = rsqrtf(q[0] / (float)D + eps);
        rstd[1] = rsqrtf(q[1] / (float)D + eps);
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        if (r == 1 && !two) break;
        T* yr = y + (int64_t)(row0 + r) * ldy;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int i = c * 512 + lane * 8;
            if (i >= D) continue;
            float o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                o[e] = RMS ? rms_out<T>(v[r][c][e], rstd[r], ww[c][e]) : ln_out(v[r][c][e], mean[r], rstd[r], ww[c][e], bb[c][e]);
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
    if (dtype == GAR_BF16 && D <= 1024 && M >= 4096) {     // D = 2048 needs 152 VGPRs and gets slower
        dim3 grid2((M + 7) / 8);
#define LAUNCH_NORM2(C_)                                                                                       \
    hipLaunchKernelGGL((norm2_kernel<bf16_t, RMS, C_>), grid2, block, 0, s, (const bf16_t*)x, (bf16_t*)y, (const bf16_t*)w, \
                       (const bf16_t*)b, M, D, ldx, ldy, eps)
        LAUNCH_NORM2(2);
#undef LAUNCH_NORM2
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

extern "C" int gar_layernorm(int dtype, const void* x, void* y, const void* w, const void* b, int M, int D, int64_t ldx,
                             int64_t ldy, float eps, gar_stream_t stream) {
    return launch_norm<false>(dtype, x, y, w, b, M, D, ldx, ldy, eps, stream);
}

extern "C" int gar_rmsnorm(int dtype, const void* x, void* y, const void* w, int M, int D, int64_t ldx, int64_t ldy,
                           float eps, gar_stream_t stream) {
    return launch_norm<true>(dtype, x, y, w, nullptr, M, D, ldx, ldy, eps, stream);
}
