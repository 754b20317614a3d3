// GEMM C[M,N] = A[M,K] * W[N,K]^T with fused epilogues — gfx950 MFMA kernels.
//
//   gemm_bf16_pp_kernel (gemm_pp.hip)  large problems: persistent 256x256x64 ping-pong kernel.
//   gemm_bf16_kernel   128x128x64 tile, 4 waves (2x2), v_mfma_f32_16x16x32_bf16, operands staged HBM->LDS by
//                      global_load_lds (16 B/lane, no VGPR round trip), LDS image XOR-swizzled through the *source*
//                      address (the DMA destination is lane-linear), double-buffered, one barrier per K tile.
//   gemm_f32_kernel    parity mode: 64x64x16 tile on v_mfma_f32_16x16x4_f32 (exact f32 fma chain).
//   skinny_*_kernel    M <= 16 (decode): weight-streaming, W fragments loaded straight to VGPRs in full 128-B lines,
//                      split-K across the waves of a block, LDS reduce, optional fused RMSNorm prologue.
//
// The MFMA operands are swapped (A-operand = W rows, B-operand = A rows) so each lane ends up holding four
// CONSECUTIVE output columns of one output row: bias/gamma/residual/stores are 8- or 16-byte vectors.
#include <stdlib.h>

#include "gemm_epilogue.h"

bool gar_gemm_pp_try(const gar_gemm_params& p, hipStream_t s);   // gemm_pp.hip
bool gar_gemm_pp_takes(const gar_gemm_params& p);                // gemm_pp.hip: the predicate alone
bool gar_skinny_bf16_try(const gar_gemm_params& p, hipStream_t s);   // gemm_skinny.hip

// ===============================================================================================================
// bf16: 128 x 128 x 64
// ===============================================================================================================
#define BM 128
#define BN 128
#define BK 64
#define TILE_BYTES (BM * BK * 2)   // 16 KiB per operand tile

// one operand tile: 128 rows x 128 B. Chunk c (8 rows = 1 KiB) is one wave-instruction; lane l lands at LDS byte
// c*1024 + l*16 = (row, slot l&7); it fetches global 16-B chunk (l&7)^(row&7) of its row  (row&7 == l>>3).
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ G, int64_t ld, int row0, int rows, int k0,
                                           char* lds, int wave, int lane) {
    const int sub = lane >> 3;
    const int c16 = (lane & 7) ^ sub;
#pragma unroll
    for (int pss = 0; pss < 4; ++pss) {
        const int c = pss * 4 + wave;
        int row = row0 + c * 8 + sub;
        row = row < rows ? row : rows - 1;
        const bf16_t* g = G + (int64_t)row * ld + k0 + c16 * 8;
        __builtin_amdgcn_global_load_lds(GLB_AS(g), LDS_AS(lds + c * 1024), 16, 0, 0);
    }
}

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(const gar_gemm_params p, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];   // [2][A 16K | W 16K]
    int tm, tn;
    tile_of(blockIdx.x, gridDim.x, tiles_m, tiles_n, tm, tn);
    const int m0 = tm * BM, n0 = tn * BN;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wave >> 1, wc = wave & 1;
    const bf16_t* A = (const bf16_t*)p.A;
    const bf16_t* W = (const bf16_t*)p.W;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int nt = p.K / BK;
    stage_tile(A, p.lda, m0, p.M, 0, smem, wave, lane);
    stage_tile(W, p.ldw, n0, p.N, 0, smem + TILE_BYTES, wave, lane);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int frow = lane & 15, fq = lane >> 4, sw = lane & 7;
    int cur = 0;
    for (int t = 0; t < nt; ++t) {
        char* As = smem + cur * (2 * TILE_BYTES);
        char* Ws = As + TILE_BYTES;
        if (t + 1 < nt) {
            char* An = smem + (cur ^ 1) * (2 * TILE_BYTES);
            stage_tile(A, p.lda, m0, p.M, (t + 1) * BK, An, wave, lane);
            stage_tile(W, p.ldw, n0, p.N, (t + 1) * BK, An + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int coff = ((kk * 4 + fq) ^ sw) << 4;
            bf16x8 af[4], wf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(As + (wr * 64 + i * 16 + frow) * 128 + coff);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                wf[j] = *reinterpret_cast<const bf16x8*>(Ws + (wc * 64 + j * 16 + frow) * 128 + coff);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = MFMA_16x16x32(wf[j], af[i], acc[i][j]);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        cur ^= 1;
    }

    // lane holds, for tile (i,j): row m = m0 + wr*64 + i*16 + (lane&15), columns n = n0 + wc*64 + j*16 + (lane>>4)*4 + r
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wr * 64 + i * 16 + frow;
        if (m >= p.M) continue;
        if (EPI == GAR_EPI_SWIGLU) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const int nin = n0 + wc * 64 + jj * 32;
                if (nin >= p.N) continue;
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu_fast(acc[i][2 * jj][r]) * acc[i][2 * jj + 1][r];
                epilogue_store<bf16_t, EPI>(p, m, (nin >> 1) + fq * 4, v);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wc * 64 + j * 16 + fq * 4;
                if (n >= p.N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue_store<bf16_t, EPI>(p, m, n, v);
            }
        }
    }
}

// ===============================================================================================================
// f32 (parity mode): 64 x 64 x 16 on v_mfma_f32_16x16x4_f32
// ===============================================================================================================
#define FBM 64
#define FBK 16
#define FLD 17

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const gar_gemm_params p, int tiles_m, int tiles_n) {
    __shared__ float As[FBM * FLD];
    __shared__ float Ws[FBM * FLD];
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const int m0 = tm * FBM, n0 = tn * FBM;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const float* A = (const float*)p.A;
    const float* W = (const float*)p.W;
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lrow = tid >> 2, lk = (tid & 3) * 4;
    const int arow = min(m0 + lrow, p.M - 1), wrow = min(n0 + lrow, p.N - 1);
    const int frow = lane & 15, fq = lane >> 4;
    for (int k0 = 0; k0 < p.K; k0 += FBK) {
        const float4 a = *reinterpret_cast<const float4*>(A + (int64_t)arow * p.lda + k0 + lk);
        const float4 w = *reinterpret_cast<const float4*>(W + (int64_t)wrow * p.ldw + k0 + lk);
        __syncthreads();
        float* ad = As + lrow * FLD + lk;
        ad[0] = a.x; ad[1] = a.y; ad[2] = a.z; ad[3] = a.w;
        float* wd = Ws + lrow * FLD + lk;
        wd[0] = w.x; wd[1] = w.y; wd[2] = w.z; wd[3] = w.w;
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float af[2], wf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = As[(wr * 32 + i * 16 + frow) * FLD + ks * 4 + fq];
#pragma unroll
            for (int j = 0; j < 2; ++j) wf[j] = Ws[(wc * 32 + j * 16 + frow) * FLD + ks * 4 + fq];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j], af[i], acc[i][j], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m0 + wr * 32 + i * 16 + frow;
        if (m >= p.M) continue;
        if (EPI == GAR_EPI_SWIGLU) {
            const int nin = n0 + wc * 32;
            if (nin < p.N) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = silu(acc[i][0][r]) * acc[i][1][r];
                epilogue_store<float, EPI>(p, m, (nin >> 1) + fq * 4, v);
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int n = n0 + wc * 32 + j * 16 + fq * 4;
                if (n >= p.N) continue;
                float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                epilogue_store<float, EPI>(p, m, n, v);
            }
        }
    }
}

// ===============================================================================================================
// skinny f32 (M <= 16, parity mode): weight streaming, split-K over the waves of a block. The bf16 decode kernel
// (M <= 64) lives in gemm_skinny.hip. NORM: fused RMSNorm prologue — the B operand is x*g (g = norm weight) and the
// accumulator is scaled by rsqrt(mean(x^2)+eps) of its row in the epilogue (rstd factors out of the dot product).
// ===============================================================================================================
// f32: lane reads float4 W[row][k0+4g..+4), x likewise; MFMA t (16x16x4) consumes element t of both.
template <int EPI, int NT, bool NORM>
__global__ __launch_bounds__(1024) void skinny_f32_kernel(const gar_gemm_params p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* red = reinterpret_cast<float*>(smem);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int frow = lane & 15, fq = lane >> 4;
    const int n0 = blockIdx.x * 16 * NT;
    const float* W = (const float*)p.W;
    const float* X = (const float*)p.A;
    const float* Gw = (const float*)p.norm_w;
    const int ksteps = p.K / 16;
    const int per = (ksteps + nw - 1) / nw;
    const int ks0 = wave * per, ks1 = min(ksteps, ks0 + per);
    const bool xvalid = frow < p.M;
    const float* xp = X + (int64_t)(xvalid ? frow : 0) * p.lda + fq * 4;
    float ssq = 0.f;
    const float* wp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) wp[t] = W + (int64_t)min(n0 + t * 16 + frow, p.N - 1) * p.ldw + fq * 4;
    f32x4 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int ks = ks0; ks < ks1; ++ks) {
        const int k0 = ks * 16;
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (xvalid) x = *reinterpret_cast<const float4*>(xp + k0);
        if (NORM) {
            const float4 g = *reinterpret_cast<const float4*>(Gw + fq * 4 + k0);
            ssq += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
            x.x *= g.x; x.y *= g.y; x.z *= g.z; x.w *= g.w;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const float4 w = *reinterpret_cast<const float4*>(wp[t] + k0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.x, x.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.y, x.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.z, x.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(w.w, x.w, acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
        *reinterpret_cast<f32x4*>(red + ((wave * NT + t) * 64 + lane) * 4) = acc[t];
    float* red_ss = red + nw * NT * 256;
    if (NORM) {
        ssq += __shfl_xor(ssq, 16, 64);
        ssq += __shfl_xor(ssq, 32, 64);
        if (lane < 16) red_ss[wave * 16 + lane] = ssq;
    }
    __syncthreads();
    if (wave != 0) return;
    float v[NT][4];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[t][r] = 0.f;
        for (int w = 0; w < nw; ++w) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(red + ((w * NT + t) * 64 + lane) * 4);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[t][r] += a[r];
        }
    }
    if (NORM) {
        float tot = 0.f;
        for (int w = 0; w < nw; ++w) tot += red_ss[w * 16 + frow];
        const float rstd = rsqrtf(tot / (float)p.K + p.norm_eps);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[t][r] *= rstd;
    }
    if (!xvalid) return;
    if (EPI == GAR_EPI_SWIGLU) {
        float o[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = silu(v[0][r]) * v[NT - 1][r];
        epilogue_store<float, EPI>(p, frow, (n0 >> 1) + fq * 4, o);
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 16 + fq * 4;
            if (n < p.N) epilogue_store<float, EPI>(p, frow, n, v[t]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// host dispatch
// ---------------------------------------------------------------------------------------------------------------
static int pick_waves(int n_blocks, int ksteps) {
    // enough waves in flight to cover HBM latency (>= ~2048 waves chip-wide) while each wave keeps >= 2 K steps
    int nw = 4;
    while (nw < 16 && n_blocks * nw < 2048 && ksteps / (nw * 2) >= 2) nw *= 2;
    return nw;
}

template <int EPI>
static int launch(int dtype, const gar_gemm_params& p, hipStream_t s) {
    if (p.row_scale || p.row_stats) {       // folded norms live in the bf16 tile GEMM's epilogues only
        if (dtype == GAR_BF16 && p.split_k <= 1 && !p.norm_w && gar_gemm_pp_try(p, s)) return GAR_OK;
        gar_set_error("gar_gemm: row_scale / row_stats are built for the bf16 tile GEMM (>= 128 tiles of 256 x 256) and the "
                      "epilogues include/gar_hip.h lists (M=%d N=%d epilogue=%d)", p.M, p.N, p.epilogue);
        return GAR_ERR_UNSUPPORTED;
    }
    if (dtype == GAR_BF16 && p.M <= 64 && gar_skinny_bf16_try(p, s)) return GAR_OK;   // gemm_skinny.hip
    if (dtype == GAR_F32 && p.M <= 16) {
        constexpr int NT = (EPI == GAR_EPI_SWIGLU) ? 2 : 1;
        const int nb = (p.N + 16 * NT - 1) / (16 * NT);
        const int nw = pick_waves(nb, p.K / 16);
        const int lds = nw * NT * 1024 + nw * 64;
        if (p.norm_w) hipLaunchKernelGGL((skinny_f32_kernel<EPI, NT, true>), dim3(nb), dim3(nw * 64), lds, s, p);
        else hipLaunchKernelGGL((skinny_f32_kernel<EPI, NT, false>), dim3(nb), dim3(nw * 64), lds, s, p);
        return GAR_OK;
    }
    if (dtype == GAR_BF16) {
        if (gar_gemm_pp_try(p, s)) return GAR_OK;
        const int tmn = (p.M + BM - 1) / BM, tnn = (p.N + BN - 1) / BN;
        hipLaunchKernelGGL((gemm_bf16_kernel<EPI>), dim3(tmn * tnn), dim3(256), 4 * TILE_BYTES, s, p, tmn, tnn);
    } else {
        const int tmn = (p.M + FBM - 1) / FBM, tnn = (p.N + FBM - 1) / FBM;
        hipLaunchKernelGGL((gemm_f32_kernel<EPI>), dim3(tmn * tnn), dim3(256), 0, s, p, tmn, tnn);
    }
    return GAR_OK;
}

// 1 when gar_gemm would run this problem on the persistent 256 x 256 bf16 tile GEMM (csrc/gemm_pp.hip) — the kernel whose
// epilogues carry the folded norms (row_scale / row_stats) and the fused qkv forms; 0 otherwise. Nothing is launched.
extern "C" int gar_gemm_tile_takes(int dtype, const gar_gemm_params* pp) {
    if (!pp || dtype != GAR_BF16 || pp->M <= 0 || pp->N <= 0 || pp->K <= 0 || pp->K % 64 != 0 || pp->split_k > 1 || pp->norm_w ||
        pp->norm_folded)
        return 0;
    if (pp->M <= 64 && !(pp->row_scale || pp->row_stats) && pp->epilogue != GAR_EPI_QKV_ROPE && pp->epilogue != GAR_EPI_QKV_ROPE_LLM &&
        pp->epilogue != GAR_EPI_PATCH_POS && pp->epilogue != GAR_EPI_BIAS_GELU && pp->epilogue != GAR_EPI_BIAS_SCALE_RES)
        return 0;                                                   // the skinny kernel is asked first for these
    return gar_gemm_pp_takes(*pp) ? 1 : 0;
}

extern "C" int gar_gemm(int dtype, const gar_gemm_params* pp, gar_stream_t stream) {
    GAR_CHECK_ARG(pp != nullptr, "gar_gemm: null params");
    const gar_gemm_params& p = *pp;
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "gar_gemm: bad dtype %d", dtype);
    GAR_CHECK_ARG(p.A && p.W && (p.C || p.epilogue == GAR_EPI_QKV_ROPE_LLM), "gar_gemm: null operand");
    GAR_CHECK_ARG(p.M > 0 && p.N > 0 && p.K > 0, "gar_gemm: bad shape %d %d %d", p.M, p.N, p.K);
    const int kq = dtype == GAR_BF16 ? 64 : 16;
    GAR_CHECK_ARG(p.K % kq == 0, "gar_gemm: K=%d must be a multiple of %d", p.K, kq);
    const int esz = dtype == GAR_BF16 ? 2 : 4;
    GAR_CHECK_ARG((p.lda * esz) % 16 == 0 && (p.ldw * esz) % 16 == 0, "gar_gemm: lda/ldw rows must be 16-byte aligned");
    GAR_CHECK_ARG(((uintptr_t)p.A % 16) == 0 && ((uintptr_t)p.W % 16) == 0, "gar_gemm: A/W must be 16-byte aligned");
    const int e = p.epilogue;
    if (p.split_k > 1)
        GAR_CHECK_ARG(dtype == GAR_BF16 && p.M <= 64 && e == GAR_EPI_NONE && !p.norm_w && p.partial && p.split_k <= 8 &&
                          p.K % (64 * p.split_k) == 0 && p.N % 4 == 0,
                      "gar_gemm: split_k=%d is built for bf16 decode GEMMs (M <= 64, GAR_EPI_NONE, K %% (64 split_k) == 0)",
                      p.split_k);
    if (e == GAR_EPI_BIAS || e == GAR_EPI_BIAS_GELU || e == GAR_EPI_BIAS_SCALE_RES)
        GAR_CHECK_ARG(p.bias != nullptr && p.N % 4 == 0, "gar_gemm: bias epilogue needs bias and N%%4==0");
    if (e == GAR_EPI_BIAS_SCALE_RES) GAR_CHECK_ARG(p.gamma && p.residual, "gar_gemm: SCALE_RES needs gamma+residual");
    if (e == GAR_EPI_RES) GAR_CHECK_ARG(p.residual && p.N % 4 == 0, "gar_gemm: RES needs residual");
    if (e == GAR_EPI_SWIGLU) GAR_CHECK_ARG(p.N % 32 == 0, "gar_gemm: SWIGLU needs N%%32==0");
    if (p.norm_w)
        GAR_CHECK_ARG(p.M <= (dtype == GAR_BF16 ? 64 : 16) &&
                          (e == GAR_EPI_NONE || e == GAR_EPI_RES || e == GAR_EPI_SWIGLU ||
                           (e == GAR_EPI_BIAS && dtype == GAR_BF16) || dtype == GAR_F32),
                      "gar_gemm: fused RMSNorm prologue is built for the decode path only (M <= 64 bf16 / 16 f32)");
    if (e == GAR_EPI_QKV_ROPE)
        GAR_CHECK_ARG(p.bias && p.qkv_q && p.qkv_k && p.qkv_sin && p.qkv_heads > 0 && p.qkv_head_dim % 8 == 0 &&
                          p.N == 3 * p.qkv_heads * p.qkv_head_dim && p.qkv_tokens > 0 && p.M % p.qkv_tokens == 0 &&
                          p.qkv_tokens_pad >= p.qkv_tokens && p.qkv_prefix >= 0 && p.qkv_prefix <= p.qkv_tokens,
                      "gar_gemm: QKV_ROPE args");
    if (e == GAR_EPI_QKV_ROPE)      // the compact (sin, cos)-pair table goes with the per-wave epilogue, which writes v head-major
        GAR_CHECK_ARG(p.qkv_cos || p.qkv_v, "gar_gemm: QKV_ROPE with the compact table (qkv_cos == NULL) needs qkv_v");
    if (p.norm_folded)
        GAR_CHECK_ARG(dtype == GAR_BF16 && p.M <= 64 && !p.norm_w && p.split_k <= 1 &&
                          (e == GAR_EPI_NONE || e == GAR_EPI_SWIGLU || e == GAR_EPI_BIAS),
                      "gar_gemm: norm_folded is built for the bf16 decode GEMMs (M <= 64; NONE / BIAS / SWIGLU)");
    if (e == GAR_EPI_QKV_ROPE_LLM) {
        const int hd_ = p.qkv_head_dim;
        GAR_CHECK_ARG(!p.bias && p.qkv_q && p.qkv_k && p.qkv_v && p.qkv_sin && p.qkv_cos && p.qkv_heads > 0 && p.qkv_kv_heads > 0 &&
                          (hd_ == 64 || hd_ == 128) && p.N == (p.qkv_heads + 2 * p.qkv_kv_heads) * hd_ && p.qkv_tokens > 0 &&
                          p.M % p.qkv_tokens == 0 && p.qkv_tokens_pad >= p.qkv_tokens && p.qkv_kv_stride > 0 && p.qkv_pos0 >= 0 &&
                          (p.qkv_pos_dev || p.qkv_pos0 + p.qkv_tokens <= p.qkv_kv_stride) &&
                          (int64_t)p.qkv_heads * p.qkv_tokens_pad * hd_ < ((int64_t)1 << 31) &&
                          (int64_t)p.qkv_kv_heads * p.qkv_kv_stride * hd_ < ((int64_t)1 << 31),
                      "gar_gemm: QKV_ROPE_LLM args");
    }
    if (e == GAR_EPI_PATCH_POS)
        GAR_CHECK_ARG(p.pos && p.tokens_in > 0 && p.tokens_out >= p.tokens_in + p.token_offset && p.N % 4 == 0,
                      "gar_gemm: PATCH_POS args");
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (e) {
        case GAR_EPI_NONE: rc = launch<GAR_EPI_NONE>(dtype, p, s); break;
        case GAR_EPI_BIAS: rc = launch<GAR_EPI_BIAS>(dtype, p, s); break;
        case GAR_EPI_BIAS_GELU: rc = launch<GAR_EPI_BIAS_GELU>(dtype, p, s); break;
        case GAR_EPI_BIAS_SCALE_RES: rc = launch<GAR_EPI_BIAS_SCALE_RES>(dtype, p, s); break;
        case GAR_EPI_RES: rc = launch<GAR_EPI_RES>(dtype, p, s); break;
        case GAR_EPI_SWIGLU: rc = launch<GAR_EPI_SWIGLU>(dtype, p, s); break;
        case GAR_EPI_PATCH_POS: rc = launch<GAR_EPI_PATCH_POS>(dtype, p, s); break;
        case GAR_EPI_QKV_ROPE:
            // fused front half of timm AttentionRope: bf16, shapes the ping-pong kernel takes; otherwise the caller
            // keeps GAR_EPI_BIAS + gar_vit_qkv_post
            if (dtype != GAR_BF16 || !gar_gemm_pp_try(p, s)) {
                gar_set_error("gar_gemm: QKV_ROPE epilogue is built for the bf16 ping-pong kernel only (M=%d N=%d)", p.M, p.N);
                return GAR_ERR_UNSUPPORTED;
            }
            rc = GAR_OK;
            break;
        case GAR_EPI_QKV_ROPE_LLM:
            // fused gar_llm_qkv_post: bf16, shapes the ping-pong kernel takes; otherwise the caller keeps GAR_EPI_NONE +
            // gar_llm_qkv_post (with W in its natural row order)
            if (dtype != GAR_BF16 || !gar_gemm_pp_try(p, s)) {
                gar_set_error("gar_gemm: QKV_ROPE_LLM epilogue is built for the bf16 ping-pong kernel only (M=%d N=%d)", p.M, p.N);
                return GAR_ERR_UNSUPPORTED;
            }
            rc = GAR_OK;
            break;
        default: gar_set_error("gar_gemm: unknown epilogue %d", e); return GAR_ERR_ARG;
    }
    if (rc != GAR_OK) return rc;
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
