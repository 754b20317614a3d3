// LayerNorm / RMSNorm — one wave per row, 8 elements (16 B bf16 / 32 B f32) per lane per step, fp32 statistics.
// HBM-bound: algorithmic bytes = 2 * M * D * sizeof(T).
#include "common.h"

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
                for (int e = 0; e < 8; ++e) s += RMS ? v[c][e] * v[c][e] : v[c][e];
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
            for (int e = 0; e < 8; ++e) s += RMS ? t[e] * t[e] : t[e];
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
                    for (int e = 0; e < 8; ++e) { const float d = v[c][e] - mean; q += d * d; }
                }
            }
        } else {
            for (int i = lane * 8; i < D; i += 512) {
                float t[8];
                ld8(xr + i, t);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = t[e] - mean; q += d * d; }
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
            for (int e = 0; e < 8; ++e) {
                float n = t[e] * rstd;
                // HF LlamaRMSNorm: weight * hidden.to(input_dtype) -> the normalised value is rounded to the storage
                // dtype before the weight multiply
                if (sizeof(T) == 2) n = bf2f(f2bf(n));
                o[e] = ww[e] * n;
            }
        } else {
            float bb[8];
            ld8(b + i, bb);
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (t[e] - mean) * rstd * ww[e] + bb[e];
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
