// Llama-side data movement kernels:
//   llm_qkv_post   half-split RoPE (llama3-scaled tables) + q scale + head-major relayout + KV-cache append
//                  (K and V both [B,Hkv,Smax,hd]: one contiguous row per appended token; the attention kernels form the PV
//                  operand with gfx950's transposing LDS read)
//   embed_lookup   embedding rows of the just-sampled tokens
//   argmax         greedy sampling with first-index tie break; writes into the device-side token matrix
//   sample         do_sample = True: temperature / top-k / top-p warpers + one Philox draw per row and step
#include "common.h"

// Column of natural dim d0 (first element of an 8-wide slice) inside a q / k head whose rows of W — and therefore whose columns
// of the qkv GEMM output — are in the fused RoPE epilogue's strip order (include/gar_hip.h, GAR_EPI_QKV_ROPE_LLM: the only
// bf16 copy of the Llama qkv weight is the folded one in that order): head_dim 128 keeps dims [0..31, 64..95, 32..63, 96..127],
// head_dim 64 the natural order.
template <int HD>
__device__ __forceinline__ int strip_col(int d0, bool strip) {
    if (HD != 128 || !strip) return d0;
    const int blk = d0 >> 5;
    return blk == 1 ? d0 + 32 : (blk == 2 ? d0 - 32 : d0);
}

// block = (batch b, 64-position chunk); loops over all heads of q, k, v.
// thread (token nl = tid/ (HD/16), i8 = 8-wide slice of the FIRST half): rotates (x[i], x[i+HD/2]) pairs of the q / k heads;
// the v heads go through the same loop unrotated (cos = 1, sin = 0) into the row-major V cache.
template <typename T, int HD>
__global__ __launch_bounds__(256) void llm_qkv_post_kernel(const T* __restrict__ qkv, const float* __restrict__ cs,
                                                           const float* __restrict__ sn, T* __restrict__ Q,
                                                           T* __restrict__ Kc, T* __restrict__ Vc, int S, int Spad,
                                                           int Hq, int Hkv, int Smax, int pos0,
                                                           const int32_t* __restrict__ pos_dev,
                                                           const int32_t* __restrict__ left_pad, float q_scale, int strip) {
    constexpr int HALF = HD / 2;
    constexpr int LPT = HALF / 8;               // lanes per token (4 for hd 64, 8 for hd 128)
    constexpr int TPP = 256 / LPT;              // tokens per pass (64 / 32)
    const int chunks = (Spad + 63) / 64;
    const int ch = blockIdx.x % chunks;
    const int b = blockIdx.x / chunks;
    const int tid = threadIdx.x;
    const int i8 = (tid % LPT) * 8;
    const int p0 = pos_dev ? pos_dev[0] : pos0;
    // left-padded batch (HF generation): sequence b starts at row left_pad[b]; the CACHE row of a token stays its row in the
    // padded sequence, only its RoPE position is counted from the first real token (rows before it: position 0, never read)
    const int lp = left_pad ? left_pad[b] : 0;
    const int nh = Hq + 2 * Hkv;
    const int W = nh * HD;
    // a thread keeps its token and its 8-wide slice for every head, so cos / sin are loaded once per pass and the rows of
    // RU heads are in flight together (a CU holds 16 waves: the latency cover has to come from here)
    constexpr int RU = 4;
    for (int pass = 0; pass < 64 / TPP; ++pass) {
        const int nl = pass * TPP + tid / LPT;
        const int s = ch * 64 + nl;
        float c[8], sv[8];
        if (s < S) {
            const int rp = max(p0 + s - lp, 0);
            ld8(cs + (int64_t)rp * HALF + i8, c);
            ld8(sn + (int64_t)rp * HALF + i8, sv);
        }
        for (int head0 = 0; head0 < nh; head0 += RU) {
            float x1[RU][8], x2[RU][8];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int head = head0 + u;
                if (s < S && head < nh) {
                    const T* row = qkv + ((int64_t)b * S + s) * W + head * HD;
                    const bool so = strip && head < Hq + Hkv;          // v heads keep the natural order
                    ld8(row + strip_col<HD>(i8, so), x1[u]);
                    ld8(row + strip_col<HD>(HALF + i8, so), x2[u]);
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) x1[u][e] = x2[u][e] = 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const int head = head0 + u;
                if (head >= nh) break;
                const bool isq = head < Hq, isv = head >= Hq + Hkv;
                if (s < S && !isv) {
                    const float sc = isq ? q_scale : 1.0f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) rope_half_pair(x1[u][e], x2[u][e], c[e], sv[e], sc, x1[u][e], x2[u][e]);
                }
                if (isq && s < Spad) {
                    T* o = Q + (((int64_t)b * Hq + head) * Spad + s) * HD;
                    st8(o + i8, x1[u]);
                    st8(o + HALF + i8, x2[u]);
                } else if (!isq && s < S) {         // k and v rows of the cache: [B, Hkv, Smax, HD] both
                    T* o = (isv ? Vc + ((int64_t)b * Hkv + (head - Hq - Hkv)) * (int64_t)Smax * HD
                                : Kc + ((int64_t)b * Hkv + (head - Hq)) * (int64_t)Smax * HD) + (int64_t)(p0 + s) * HD;
                    st8(o + i8, x1[u]);
                    st8(o + HALF + i8, x2[u]);
                }
            }
        }
    }
}

// decode (S == 1): one thread per (batch, head, 8-wide slice of the first half) — a single small launch.
template <typename T, int HD>
__global__ __launch_bounds__(256) void llm_qkv_post_decode_kernel(const T* __restrict__ qkv, const float* __restrict__ cs,
                                                                  const float* __restrict__ sn, T* __restrict__ Q,
                                                                  T* __restrict__ Kc, T* __restrict__ Vc, int B, int Spad,
                                                                  int Hq, int Hkv, int Smax, int pos0,
                                                                  const int32_t* __restrict__ pos_dev,
                                                                  const int32_t* __restrict__ left_pad, float q_scale, int strip) {
    constexpr int HALF = HD / 2, LPH = HALF / 8;
    const int nh = Hq + 2 * Hkv;
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * nh * LPH) return;
    const int i8 = (idx % LPH) * 8;
    const int head = (idx / LPH) % nh;
    const int b = idx / (LPH * nh);
    const int p0 = pos_dev ? pos_dev[0] : pos0;
    const T* row = qkv + (int64_t)b * nh * HD + head * HD;
    float x1[8], x2[8];
    const bool so = strip && head < Hq + Hkv;
    ld8(row + strip_col<HD>(i8, so), x1);
    ld8(row + strip_col<HD>(HALF + i8, so), x2);
    if (head < Hq + Hkv) {
        const int rp = max(p0 - (left_pad ? left_pad[b] : 0), 0);          // RoPE position; the cache row stays p0
        const float* cp = cs + (int64_t)rp * HALF + i8;
        const float* sp = sn + (int64_t)rp * HALF + i8;
        const float sc = head < Hq ? q_scale : 1.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) rope_half_pair(x1[e], x2[e], cp[e], sp[e], sc, x1[e], x2[e]);
        T* o = head < Hq ? Q + (((int64_t)b * Hq + head) * Spad) * HD
                         : Kc + (((int64_t)b * Hkv + (head - Hq)) * Smax + p0) * HD;
        st8(o + i8, x1);
        st8(o + HALF + i8, x2);
    } else {
        T* o = Vc + (((int64_t)b * Hkv + (head - Hq - Hkv)) * Smax + p0) * HD;
        st8(o + i8, x1);
        st8(o + HALF + i8, x2);
    }
}

extern "C" int gar_llm_qkv_post(int dtype, const void* qkv, const float* cs, const float* sn, void* Q, void* Kc,
                                void* Vc, int B, int S, int Spad, int Hq, int Hkv, int hd, int Smax, int pos0,
                                const int32_t* pos_dev, const int32_t* left_pad, float q_scale, int qk_strip_order,
                                gar_stream_t stream) {
    GAR_CHECK_ARG(qkv && cs && sn && Q && Kc && Vc, "llm_qkv_post: null pointer");
    const int strip = qk_strip_order ? 1 : 0;
    GAR_CHECK_ARG(B > 0 && S > 0 && Spad >= S && Smax % 64 == 0, "llm_qkv_post: bad shape");
    GAR_CHECK_ARG(pos_dev || pos0 + S <= Smax, "llm_qkv_post: cache overflow %d+%d > %d", pos0, S, Smax);
    GAR_CHECK_ARG(hd == 64 || hd == 128, "llm_qkv_post: head_dim %d not built (64, 128)", hd);
    hipStream_t s = (hipStream_t)stream;
    if (S == 1) {
        const int total = B * (Hq + 2 * Hkv) * (hd / 16);
        dim3 g1((total + 255) / 256), b1(256);
#define LAUNCH_LQD(TT, HD_)                                                                                         \
    hipLaunchKernelGGL((llm_qkv_post_decode_kernel<TT, HD_>), g1, b1, 0, s, (const TT*)qkv, cs, sn, (TT*)Q, (TT*)Kc, \
                       (TT*)Vc, B, Spad, Hq, Hkv, Smax, pos0, pos_dev, left_pad, q_scale, strip)
        if (dtype == GAR_BF16) { if (hd == 64) LAUNCH_LQD(bf16_t, 64); else LAUNCH_LQD(bf16_t, 128); }
        else { if (hd == 64) LAUNCH_LQD(float, 64); else LAUNCH_LQD(float, 128); }
#undef LAUNCH_LQD
        GAR_CHECK_LAUNCH();
        return GAR_OK;
    }
    dim3 grid(B * ((Spad + 63) / 64)), block(256);
#define LAUNCH_LQP(TT, HD_)                                                                                       \
    hipLaunchKernelGGL((llm_qkv_post_kernel<TT, HD_>), grid, block, 0, s, (const TT*)qkv, cs, sn, (TT*)Q, (TT*)Kc,  \
                       (TT*)Vc, S, Spad, Hq, Hkv, Smax, pos0, pos_dev, left_pad, q_scale, strip)
    if (dtype == GAR_BF16) { if (hd == 64) LAUNCH_LQP(bf16_t, 64); else LAUNCH_LQP(bf16_t, 128); }
    else { if (hd == 64) LAUNCH_LQP(float, 64); else LAUNCH_LQP(float, 128); }
#undef LAUNCH_LQP
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void embed_lookup_kernel(const int64_t* __restrict__ tokens, const T* __restrict__ E,
                                                           T* __restrict__ out, int C, int64_t vocab) {
    int64_t id = tokens[blockIdx.x];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const T* src = E + id * C;
    T* dst = out + (int64_t)blockIdx.x * C;
    for (int i = threadIdx.x * 8; i < C; i += 256 * 8) {
        float v[8];
        ld8(src + i, v);
        st8(dst + i, v);
    }
}

extern "C" int gar_embed_lookup(int dtype, const int64_t* tokens, const void* E, void* out, int B, int C, int64_t vocab,
                                gar_stream_t stream) {
    GAR_CHECK_ARG(tokens && E && out && B > 0 && C % 8 == 0, "embed_lookup: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((embed_lookup_kernel<bf16_t>), dim3(B), dim3(256), 0, s, tokens, (const bf16_t*)E, (bf16_t*)out,
                           C, vocab);
    else
        hipLaunchKernelGGL((embed_lookup_kernel<float>), dim3(B), dim3(256), 0, s, tokens, (const float*)E, (float*)out, C,
                           vocab);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// argmax: stage 1 = AM_BLOCKS partial (max, first index) per row; stage 2 = one wave per row.
// ---------------------------------------------------------------------------------------------------------------
#define AM_BLOCKS 64

__device__ __forceinline__ void am_better(float& bv, int& bi, float v, int i) {
    if (v > bv || (v == bv && i < bi)) { bv = v; bi = i; }
}

template <typename T>
__global__ __launch_bounds__(256) void argmax_partial_kernel(const T* __restrict__ logits, int64_t ld, int V,
                                                             float* __restrict__ pv, int* __restrict__ pi) {
    __shared__ float sv[4];
    __shared__ int si[4];
    const int b = blockIdx.y, blk = blockIdx.x;
    const T* row = logits + (int64_t)b * ld;
    const int per = (V + AM_BLOCKS - 1) / AM_BLOCKS;
    const int lo = blk * per, hi = min(V, lo + per);
    float bv = -INFINITY;
    int bi = 0x7fffffff;
    for (int i = lo + threadIdx.x; i < hi; i += 256) am_better(bv, bi, DT<T>::ld(row + i), i);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        am_better(bv, bi, ov, oi);
    }
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = bv; si[threadIdx.x >> 6] = bi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 4; ++w) am_better(bv, bi, sv[w], si[w]);
        pv[b * AM_BLOCKS + blk] = bv;
        pi[b * AM_BLOCKS + blk] = bi;
    }
}

__global__ __launch_bounds__(64) void argmax_final_kernel(const float* __restrict__ pv, const int* __restrict__ pi,
                                                          int64_t* __restrict__ out_tokens, int64_t out_stride,
                                                          const int32_t* __restrict__ step_dev,
                                                          int64_t* __restrict__ cur_tokens,
                                                          const int64_t* __restrict__ eos_ids, int n_eos,
                                                          int32_t* __restrict__ finished, int32_t* __restrict__ done_count) {
    const int b = blockIdx.x;
    float bv = pv[b * AM_BLOCKS + threadIdx.x];
    int bi = pi[b * AM_BLOCKS + threadIdx.x];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(bv, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        am_better(bv, bi, ov, oi);
    }
    if (threadIdx.x == 0) {
        const int step = step_dev ? step_dev[0] : 0;
        if (out_tokens) out_tokens[(int64_t)b * out_stride + step] = bi;
        if (cur_tokens) cur_tokens[b] = bi;
        // greedy stopping criterion on the device (HF: EosTokenCriteria over eos_token_id, a list): the FIRST step at which row b
        // produced an end-of-sequence id is latched in finished[b] (-1 = still running), and done_count counts the latched rows —
        // the host's "is every row done" poll reads one int instead of scanning the token matrix (entries < 0 of eos_ids never match)
        if (finished && eos_ids && finished[b] < 0) {
            bool hit = false;
            for (int e = 0; e < n_eos; ++e) hit |= (eos_ids[e] == (int64_t)bi);
            if (hit) {
                finished[b] = step;
                if (done_count) atomicAdd(done_count, 1);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Sampling (do_sample = True): HF's warpers in their order — temperature, top-k, top-p — then one draw from what is left
// (transformers TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper + torch.multinomial in GenerationMixin._sample;
// the reference forwards the caller's GenerationConfig, modeling_gar.py:418-426). One 1024-thread workgroup per row, the row read
// ~12 times out of L2 (128 k logits: tens of microseconds); no sort:
//   z = logit / T (fp32 division), ordered by an order-preserving 32-bit key;
//   top-k: the k-th largest key by a 4 x 8-bit radix descent over COUNTS; keep key >= it (ties at the k-th value stay, as
//          `scores < topk(scores, k)[0][..., -1]` removes strictly smaller ones only);
//   top-p: a token stays iff the probability mass of the strictly larger keys is < top_p (HF removes ascending-cumsum <= 1 - top_p:
//          the same set up to ties, which stay or go together here) — the smallest such key by the same descent over MASS;
//   draw: u = 24 random bits of Philox4x32-10(key = seed, counter = (step, row)) / 2^24; the token is the first index (in VOCABULARY
//          order) whose running sum of kept exp(z - max) exceeds u x their total. Deterministic: per-wave histograms merged in wave
//          order, fixed-shape scans — the same (logits, seed, step, row) always give the same token; oracle/sampling.py restates it.
// ---------------------------------------------------------------------------------------------------------------
#define SMP_THREADS 1024
#define SMP_WAVES (SMP_THREADS / 64)

__device__ __forceinline__ unsigned int smp_key(float z) {           // larger z <-> larger key; -0 and +0 differ (harmless), NaN sorts high
    const unsigned int u = __float_as_uint(z);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

__device__ __forceinline__ unsigned int smp_mulhi(unsigned int a, unsigned int b) { return __umulhi(a, b); }

// Philox4x32-10 (Salmon et al. 2011), first output word
__device__ __forceinline__ unsigned int philox_first(unsigned int c0, unsigned int c1, unsigned int c2, unsigned int c3, unsigned int k0,
                                                     unsigned int k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned int hi0 = smp_mulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const unsigned int hi1 = smp_mulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const unsigned int n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c0;
}

template <typename T>
__global__ __launch_bounds__(SMP_THREADS) void sample_kernel(const T* __restrict__ logits, int64_t ld, int V,
                                                             int64_t* __restrict__ out_tokens, int64_t out_stride,
                                                             const int32_t* __restrict__ step_dev, int64_t* __restrict__ cur_tokens,
                                                             const float* __restrict__ params, const int64_t* __restrict__ seed_dev,
                                                             const int64_t* __restrict__ eos_ids, int n_eos,
                                                             int32_t* __restrict__ finished, int32_t* __restrict__ done_count,
                                                             int row_offset) {
    __shared__ float hist_f[SMP_WAVES][256];        // per-wave mass histograms (merged in wave order: deterministic)
    __shared__ int hist_i[256];
    __shared__ float red_f[SMP_WAVES];
    __shared__ float scan_f[SMP_THREADS];
    __shared__ unsigned int sh_u[4];
    __shared__ float sh_f[4];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const T* row = logits + (int64_t)b * ld;
    const float temperature = params[0], top_p = params[1];
    const int top_k = (int)params[2];
    const float temp = temperature > 0.f ? temperature : 1.0f;             // (the host refuses temperature <= 0, as HF does)
    auto zof = [&](int i) { return DT<T>::ld(row + i) / temp; };
    auto block_sum = [&](float v) -> float {        // fixed shape: xor tree inside a wave, waves in order
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        __syncthreads();
        if (lane == 0) red_f[wave] = v;
        __syncthreads();
        float t = 0.f;
        for (int w = 0; w < SMP_WAVES; ++w) t += red_f[w];
        return t;
    };
    // ---- max
    float mx = -INFINITY;
    for (int i = tid; i < V; i += SMP_THREADS) mx = fmaxf(mx, zof(i));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red_f[wave] = mx;
    __syncthreads();
    mx = red_f[0];
    for (int w = 1; w < SMP_WAVES; ++w) mx = fmaxf(mx, red_f[w]);
    __syncthreads();
    // ---- top-k: the k-th largest key
    unsigned int key_lo = 0u;                       // keep key >= key_lo
    if (top_k > 0 && top_k < V) {
        unsigned int prefix = 0u;
        int want = top_k;                           // rank (from the top) still to be found inside the current prefix
        for (int level = 3; level >= 0; --level) {
            const int sh = level * 8;
            const unsigned int pmask = level == 3 ? 0u : (0xffffffffu << (sh + 8));
            for (int i = tid; i < 256; i += SMP_THREADS) hist_i[i] = 0;
            __syncthreads();
            for (int i = tid; i < V; i += SMP_THREADS) {
                const unsigned int k = smp_key(zof(i));
                if ((k & pmask) == prefix) atomicAdd(&hist_i[(k >> sh) & 255u], 1);
            }
            __syncthreads();
            if (tid == 0) {
                int bsel = 0, acc = 0;
                for (int q = 255; q >= 0; --q) {
                    if (acc + hist_i[q] >= want) { bsel = q; break; }
                    acc += hist_i[q];
                }
                sh_u[0] = (unsigned int)bsel;
                sh_u[1] = (unsigned int)(want - acc);
            }
            __syncthreads();
            prefix |= sh_u[0] << sh;
            want = (int)sh_u[1];
            __syncthreads();
        }
        key_lo = prefix;
    }
    // ---- mass of what top-k left
    float part = 0.f;
    for (int i = tid; i < V; i += SMP_THREADS) {
        const float z = zof(i);
        part += smp_key(z) >= key_lo ? __expf(z - mx) : 0.f;
    }
    const float Z1 = block_sum(part);
    // ---- top-p: the smallest key whose strictly-larger keys hold less than top_p of the mass
    if (top_p < 1.0f) {
        const float lim = top_p * Z1;
        unsigned int prefix = 0u;
        float above = 0.f;
        for (int level = 3; level >= 0; --level) {
            const int sh = level * 8;
            const unsigned int pmask = level == 3 ? 0u : (0xffffffffu << (sh + 8));
            for (int i = tid; i < SMP_WAVES * 256; i += SMP_THREADS) (&hist_f[0][0])[i] = 0.f;
            for (int i = tid; i < 256; i += SMP_THREADS) hist_i[i] = 0;
            __syncthreads();
            for (int i0 = wave * 64; i0 < V; i0 += SMP_THREADS) {          // a wave's lanes add in lane order inside one DS instruction
                const int i = i0 + lane;
                if (i < V) {
                    const float z = zof(i);
                    const unsigned int k = smp_key(z);
                    if (k >= key_lo && (k & pmask) == prefix) {
                        atomicAdd(&hist_f[wave][(k >> sh) & 255u], __expf(z - mx));
                        atomicAdd(&hist_i[(k >> sh) & 255u], 1);
                    }
                }
            }
            __syncthreads();
            if (tid < 256) {
                float t = 0.f;
                for (int w = 0; w < SMP_WAVES; ++w) t += hist_f[w][tid];
                scan_f[tid] = t;
            }
            __syncthreads();
            if (tid == 0) {
                // lowest non-empty bucket whose mass-above is still < lim (the top bucket always qualifies: above = 0 at level 3)
                int bsel = -1;
                float run = above, run_sel = above;
                for (int q = 255; q >= 0; --q) {
                    if (hist_i[q] > 0) {
                        if (run < lim || bsel < 0) { bsel = q; run_sel = run; }
                        else break;
                    }
                    run += scan_f[q];
                }
                sh_u[0] = (unsigned int)max(bsel, 0);
                sh_f[0] = run_sel;
            }
            __syncthreads();
            prefix |= sh_u[0] << sh;
            above = sh_f[0];
            __syncthreads();
        }
        key_lo = max(key_lo, prefix);
    }
    // ---- draw: contiguous chunks of the vocabulary per thread, exclusive scan of the chunk sums, the owning thread walks its chunk
    const int per = (V + SMP_THREADS - 1) / SMP_THREADS;
    const int c0 = tid * per, c1 = min(V, c0 + per);
    float csum = 0.f;
    for (int i = c0; i < c1; ++i) {
        const float z = zof(i);
        csum += smp_key(z) >= key_lo ? __expf(z - mx) : 0.f;
    }
    scan_f[tid] = csum;
    __syncthreads();
    if (tid < SMP_WAVES) {                          // 16 segment sums of 64 chunks each, sequential: a fixed order
        float t = 0.f;
        for (int j = 0; j < 64; ++j) t += scan_f[tid * 64 + j];
        red_f[tid] = t;
    }
    __syncthreads();
    const int step = step_dev ? step_dev[0] : 0;
    if (tid == 0) {
        float Z2 = 0.f;
        for (int w = 0; w < SMP_WAVES; ++w) Z2 += red_f[w];
        const unsigned long long seed = (unsigned long long)seed_dev[0];
        const unsigned int r = philox_first((unsigned int)step, (unsigned int)(b + row_offset), 0u, 0u, (unsigned int)seed, (unsigned int)(seed >> 32));
        const float u = (float)(r >> 8) * (1.0f / 16777216.0f);
        const float target = u * Z2;
        // segment, then chunk, then (below) element: running sums in the same fixed order
        float run = 0.f;
        int seg = SMP_WAVES - 1;
        for (int w = 0; w < SMP_WAVES; ++w) {
            if (run + red_f[w] > target) { seg = w; break; }
            run += red_f[w];
        }
        if (seg == SMP_WAVES - 1 && !(run + red_f[seg] > target)) {       // rounding at the very end: fall back to the whole last segment
            run = 0.f;
            for (int w = 0; w < seg; ++w) run += red_f[w];
        }
        int ch = seg * 64 + 63, lastnz = -1;
        float run_nz = run;
        bool found = false;
        for (int j = 0; j < 64; ++j) {
            const float cs = scan_f[seg * 64 + j];
            if (run + cs > target) { ch = seg * 64 + j; found = true; break; }
            if (cs > 0.f) { lastnz = seg * 64 + j; run_nz = run; }
            run += cs;
        }
        if (!found && lastnz >= 0) {      // the chunk-by-chunk sum rounded below the segment sum: the segment's last kept element
            ch = lastnz;
            run = run_nz;
            found = true;
        }
        sh_u[0] = (unsigned int)ch;
        sh_u[1] = found ? 1u : 0u;
        sh_f[0] = run;
        sh_f[1] = target;
    }
    __syncthreads();
    if (tid == (int)sh_u[0] || (!sh_u[1] && tid == 0)) {
        int tok = -1;
        if (sh_u[1] && tid == (int)sh_u[0]) {
            float run = sh_f[0];
            const float target = sh_f[1];
            for (int i = c0; i < c1; ++i) {
                const float z = zof(i);
                if (smp_key(z) >= key_lo) {
                    run += __expf(z - mx);
                    tok = i;                               // the last kept index seen: the answer if rounding leaves run == target
                    if (run > target) break;
                }
            }
        }
        if (tok < 0 && tid == 0) {                         // nothing exceeded the target (u x total rounded up to the total): last kept index
            for (int i = V - 1; i >= 0; --i)
                if (smp_key(zof(i)) >= key_lo) { tok = i; break; }
        }
        if (tok >= 0) {
            if (out_tokens) out_tokens[(int64_t)b * out_stride + step] = tok;
            if (cur_tokens) cur_tokens[b] = tok;
            if (finished && eos_ids && finished[b] < 0) {
                bool hit = false;
                for (int e = 0; e < n_eos; ++e) hit |= (eos_ids[e] == (int64_t)tok);
                if (hit) {
                    finished[b] = step;
                    if (done_count) atomicAdd(done_count, 1);
                }
            }
        }
    }
}

extern "C" int gar_sample(int dtype, const void* logits, int64_t ld, int B, int V, int64_t* out_tokens, int64_t out_stride,
                          const int32_t* step_dev, int64_t* cur_tokens, const float* params_dev, const int64_t* seed_dev,
                          const int64_t* eos_ids, int n_eos, int32_t* finished, int32_t* done_count, int row_offset,
                          gar_stream_t stream) {
    GAR_CHECK_ARG(logits && params_dev && seed_dev && B > 0 && V > 0 && (out_tokens || cur_tokens) && row_offset >= 0, "sample: bad args");
    GAR_CHECK_ARG(n_eos >= 0 && (n_eos == 0 || eos_ids) && (!finished || eos_ids), "sample: eos_ids / finished");
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((sample_kernel<bf16_t>), dim3(B), dim3(SMP_THREADS), 0, s, (const bf16_t*)logits, ld, V, out_tokens, out_stride,
                           step_dev, cur_tokens, params_dev, seed_dev, eos_ids, n_eos, finished, done_count, row_offset);
    else
        hipLaunchKernelGGL((sample_kernel<float>), dim3(B), dim3(SMP_THREADS), 0, s, (const float*)logits, ld, V, out_tokens, out_stride,
                           step_dev, cur_tokens, params_dev, seed_dev, eos_ids, n_eos, finished, done_count, row_offset);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

extern "C" int64_t gar_argmax_workspace(int B, int V) {
    (void)V;
    return (int64_t)B * AM_BLOCKS * 8;
}

extern "C" int gar_argmax(int dtype, const void* logits, int64_t ld, int B, int V, int64_t* out_tokens,
                          int64_t out_stride, const int32_t* step_dev, int64_t* cur_tokens, void* workspace,
                          const int64_t* eos_ids, int n_eos, int32_t* finished, int32_t* done_count, gar_stream_t stream) {
    GAR_CHECK_ARG(logits && workspace && B > 0 && V > 0 && (out_tokens || cur_tokens), "argmax: bad args");
    GAR_CHECK_ARG(n_eos >= 0 && (n_eos == 0 || eos_ids) && (!finished || eos_ids), "argmax: eos_ids / finished");
    float* pv = (float*)workspace;
    int* pi = (int*)(pv + (int64_t)B * AM_BLOCKS);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid(AM_BLOCKS, B);
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((argmax_partial_kernel<bf16_t>), grid, dim3(256), 0, s, (const bf16_t*)logits, ld, V, pv, pi);
    else
        hipLaunchKernelGGL((argmax_partial_kernel<float>), grid, dim3(256), 0, s, (const float*)logits, ld, V, pv, pi);
    hipLaunchKernelGGL(argmax_final_kernel, dim3(B), dim3(64), 0, s, pv, pi, out_tokens, out_stride, step_dev, cur_tokens, eos_ids,
                       n_eos, finished, done_count);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
