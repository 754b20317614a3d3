// Epilogues and tile mapping shared by the GEMM kernels.
#pragma once
#include "common.h"

#define LDS_AS(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_AS(p) ((const __attribute__((address_space(1))) void*)(p))

// `v` = 4 accumulators of row m, columns n..n+3 (n % 4 == 0)
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store(const gar_gemm_params& p, int m, int n, float (&v)[4]) {
    T* C = (T*)p.C;
    if (EPI == GAR_EPI_BIAS || EPI == GAR_EPI_BIAS_GELU || EPI == GAR_EPI_BIAS_SCALE_RES) {
        float b[4];
        ld4((const T*)p.bias + n, b);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += b[r];
    }
    if (EPI == GAR_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = sizeof(T) == 2 ? gelu_fast(v[r]) : gelu_erf(v[r]);
    }
    if (EPI == GAR_EPI_BIAS_SCALE_RES) {
        float g[4], res[4];
        ld4((const T*)p.gamma + n, g);
        ld4((const T*)p.residual + (int64_t)m * p.ldr + n, res);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = res[r] + g[r] * v[r];
    }
    if (EPI == GAR_EPI_RES) {
        float res[4];
        ld4((const T*)p.residual + (int64_t)m * p.ldr + n, res);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = res[r] + v[r];
    }
    int64_t off;
    if (EPI == GAR_EPI_PATCH_POS) {
        int tile = m / p.tokens_in;
        int tok = p.token_offset + (m - tile * p.tokens_in);
        float pe[4];
        ld4((const T*)p.pos + (int64_t)tok * p.N + n, pe);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] += pe[r];
        off = ((int64_t)tile * p.tokens_out + tok) * p.ldc + n;
    } else {
        off = (int64_t)m * p.ldc + n;
    }
    const int ncols = (EPI == GAR_EPI_SWIGLU) ? (p.N >> 1) : p.N;
    if (n + 3 < ncols && ((off * (int64_t)sizeof(T)) & (sizeof(T) * 4 - 1)) == 0) {
        st4(C + off, v);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (n + r < ncols) DT<T>::st(C + off + r, v[r]);
    }
}

// 8 consecutive columns (n % 8 == 0): 16-byte bias / gamma / residual loads and one 16-byte store when aligned,
// otherwise two 4-wide pieces. Not used with SWIGLU.
template <typename T, int EPI>
__device__ __forceinline__ void epilogue_store8(const gar_gemm_params& p, int m, int n, float (&v)[8]) {
    int64_t off;
    int tok = 0;
    if (EPI == GAR_EPI_PATCH_POS) {
        const int tile = m / p.tokens_in;
        tok = p.token_offset + (m - tile * p.tokens_in);
        off = ((int64_t)tile * p.tokens_out + tok) * p.ldc + n;
    } else {
        off = (int64_t)m * p.ldc + n;
    }
    const bool fast = n + 7 < p.N && ((off * (int64_t)sizeof(T)) & (sizeof(T) * 8 - 1)) == 0 &&
                      ((EPI != GAR_EPI_BIAS_SCALE_RES && EPI != GAR_EPI_RES) ||
                       ((((int64_t)m * p.ldr + n) * (int64_t)sizeof(T)) & (sizeof(T) * 8 - 1)) == 0);
    if (!fast) {
        float a[4] = {v[0], v[1], v[2], v[3]}, b[4] = {v[4], v[5], v[6], v[7]};
        if (n < p.N) epilogue_store<T, EPI>(p, m, n, a);
        if (n + 4 < p.N) epilogue_store<T, EPI>(p, m, n + 4, b);
        return;
    }
    if (EPI == GAR_EPI_BIAS || EPI == GAR_EPI_BIAS_GELU || EPI == GAR_EPI_BIAS_SCALE_RES) {
        float b[8];
        ld8((const T*)p.bias + n, b);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += b[r];
    }
    if (EPI == GAR_EPI_BIAS_GELU) {
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = sizeof(T) == 2 ? gelu_fast(v[r]) : gelu_erf(v[r]);
    }
    if (EPI == GAR_EPI_BIAS_SCALE_RES) {
        float g[8], res[8];
        ld8((const T*)p.gamma + n, g);
        ld8((const T*)p.residual + (int64_t)m * p.ldr + n, res);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = res[r] + g[r] * v[r];
    }
    if (EPI == GAR_EPI_RES) {
        float res[8];
        ld8((const T*)p.residual + (int64_t)m * p.ldr + n, res);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = res[r] + v[r];
    }
    if (EPI == GAR_EPI_PATCH_POS) {
        float pe[8];
        ld8((const T*)p.pos + (int64_t)tok * p.N + n, pe);
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] += pe[r];
    }
    st8((T*)p.C + off, v);
}

// XCD-aware tile order: the dispatcher places block b on XCD b%8; give every XCD a contiguous run of tiles (bijective
// for any count) and walk the run in groups of GM row-panels so neighbouring blocks share A/W panels in that L2.
// `v` = virtual block id in [0, nwg). GM m-tiles per group: 32 consecutive tiles of an XCD are GM x (32 / GM) tiles.
__device__ __forceinline__ void tile_of(int v, int nwg, int tiles_m, int tiles_n, int& tm, int& tn, int GM = 8) {
    const int q = nwg >> 3, r = nwg & 7, xcd = v & 7;
    const int wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (v >> 3);
    const int gsz = GM * tiles_n;
    const int g = wg / gsz;
    const int first_m = g * GM;
    const int gm = min(tiles_m - first_m, GM);
    const int in = wg - g * gsz;
    tm = first_m + in % gm;
    tn = in / gm;
}

// m-tiles per group for the persistent 256 x 256 kernel: the A panels a group keeps re-reading (GM x K x 512 B) should
// stay near the 4 MiB of an XCD's L2 — 8 panels at K = 1024, 4 at 2048, 2 at 4096, 1 from 8192 (same-box sweeps of
// -DGAR_TILE_GM builds on the benchmark's shapes, profiles/r2_gemm_tile_group.txt: gate/up -3.4 ... -5.6 %, down
// -2 ... -2.8 %, fc2 -0.8 % against 8 everywhere; 16 and 32 lose 4 - 30 % at K >= 4096)
// Narrow outputs (<= 8 n-tiles) at K <= 2048 keep 8: Llama o-proj measures 0.81 ms with 8 (or 16) against 0.83 with 4.
__device__ __forceinline__ int tile_group_m(int K, int tiles_n) {
#ifdef GAR_TILE_GM
    return GAR_TILE_GM;
#else
    // round 5 re-sweep on the final epilogues (profiles/r5_gemm_epilogue_traffic.txt 7): the ViT proj (K = 1024, 4 n-tiles) takes 16
    // m-panels per group (0.795 -> 0.778 ms), Llama's qkv (K = 2048, 12 n-tiles) 8 like o-proj (1.233 -> 1.214 ms)
    if (K <= 1024 && tiles_n <= 4) return 16;
    if (K <= 2048 && tiles_n <= 12) return 8;
    return max(1, min(8, 8192 / K));
#endif
}
