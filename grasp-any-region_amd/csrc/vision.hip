// Vision-side data movement kernels (all HBM-bound):
//   patch_im2col   mask decode + 14x14 patch gather of pixel_values and the binary mask into one GEMM operand
//   cls_pos_fill   cls token row
//   vit_qkv_post   2-D interleaved RoPE + q scale + head-major relayout (+ V transposed through LDS)
//   pool2x2        PerceptionLMAdaptiveAvgPooling (exact 2x2 mean over the token grid)
#include "common.h"

// ---------------------------------------------------------------------------------------------------------------
// mask decode in the arithmetic of the storage dtype (reference: modeling_gar.py:315-327 evaluates
// round((m + 1.0) / 2.0 * 255.0) in the tensor's dtype, so bf16 inputs round after every op)
// ---------------------------------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float rnd(float v);
template <> __device__ __forceinline__ float rnd<float>(float v) { return v; }
template <> __device__ __forceinline__ float rnd<bf16_t>(float v) { return bf2f(f2bf(v)); }

template <typename T>
__device__ __forceinline__ float mask_binary(float m, int P) {
    float t = rnd<T>(m + 1.0f);
    t = rnd<T>(t / 2.0f);
    t = rnd<T>(t * 255.0f);
    long long v = (long long)rintf(t);          // torch.round = half to even; .long()
    v = v < 0 ? 0 : (v > P ? P : v);
    return v != P ? 1.0f : 0.0f;
}

// one thread per (patch row r of out, channel c, ky): copies `patch` pixels and `patch` mask values.
template <typename T>
__global__ __launch_bounds__(256) void patch_im2col_kernel(const T* __restrict__ pixel, const T* __restrict__ mask,
                                                           T* __restrict__ out, int T_, int img, int patch, int Kp,
                                                           int P) {
    const int g = img / patch;
    const int pp = patch * patch;
    const int64_t total = (int64_t)T_ * g * g * 3 * patch;
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ky = (int)(idx % patch);
    const int c = (int)((idx / patch) % 3);
    const int64_t r = idx / (3 * patch);               // out row = (tile, py, px)
    const int px = (int)(r % g), py = (int)((r / g) % g), t = (int)(r / ((int64_t)g * g));
    const int64_t src = (((int64_t)t * 3 + c) * img + (py * patch + ky)) * img + px * patch;
    T* o = out + r * Kp + c * pp + ky * patch;
    for (int kx = 0; kx < patch; ++kx) o[kx] = pixel[src + kx];
    T* om = o + 3 * pp;
    if (mask) {
        for (int kx = 0; kx < patch; ++kx) DT<T>::st(om + kx, mask_binary<T>(DT<T>::ld(mask + src + kx), P));
    } else {
        for (int kx = 0; kx < patch; ++kx) DT<T>::st(om + kx, 0.f);
    }
    if (c == 0 && ky == 0)
        for (int k = 6 * pp; k < Kp; ++k) DT<T>::st(out + r * Kp + k, 0.f);
}

extern "C" int gar_patch_im2col(int dtype, const void* pixel, const void* mask, void* out, int T_, int img, int patch,
                                int Kp, int prompt_numbers, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "patch_im2col: bad dtype");
    GAR_CHECK_ARG(pixel && out && T_ > 0 && img > 0 && patch > 0 && img % patch == 0, "patch_im2col: bad args");
    GAR_CHECK_ARG(Kp >= 6 * patch * patch, "patch_im2col: Kp=%d < 6*patch^2", Kp);
    const int g = img / patch;
    const int64_t total = (int64_t)T_ * g * g * 3 * patch;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((patch_im2col_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)pixel, (const bf16_t*)mask,
                           (bf16_t*)out, T_, img, patch, Kp, prompt_numbers);
    else
        hipLaunchKernelGGL((patch_im2col_kernel<float>), grid, block, 0, s, (const float*)pixel, (const float*)mask,
                           (float*)out, T_, img, patch, Kp, prompt_numbers);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// mask decode alone (A1), for gar_patch_embed: 8 elements per thread, 16-byte accesses (bf16) — reads the processor's
// [-1, 1] encoding once, writes the {0, 1} mask once.
template <typename T>
__global__ __launch_bounds__(256) void mask_decode_kernel(const T* __restrict__ mask, T* __restrict__ out, int64_t n8, int P) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n8) return;
    float v[8];
    ld8(mask + i * 8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = mask_binary<T>(v[e], P);
    st8(out + i * 8, v);
}

extern "C" int gar_mask_decode(int dtype, const void* mask, void* out, int64_t n, int prompt_numbers, gar_stream_t stream) {
    GAR_CHECK_ARG(dtype == GAR_F32 || dtype == GAR_BF16, "mask_decode: bad dtype");
    GAR_CHECK_ARG(mask && out && n > 0 && n % 8 == 0, "mask_decode: n=%lld must be a positive multiple of 8", (long long)n);
    GAR_CHECK_ARG(((uintptr_t)mask % 16) == 0 && ((uintptr_t)out % 16) == 0, "mask_decode: 16-byte aligned pointers");
    const int64_t n8 = n / 8;
    dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((mask_decode_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)mask, (bf16_t*)out, n8, prompt_numbers);
    else
        hipLaunchKernelGGL((mask_decode_kernel<float>), grid, block, 0, s, (const float*)mask, (float*)out, n8, prompt_numbers);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ void cls_pos_fill_kernel(T* x, const T* cls, const T* pos, int T_, int tokens, int D) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= T_ * D) return;
    const int t = i / D, d = i % D;
    DT<T>::st(x + (int64_t)t * tokens * D + d, DT<T>::ld(cls + d) + DT<T>::ld(pos + d));
}

extern "C" int gar_cls_pos_fill(int dtype, void* x, const void* cls, const void* pos, int T_, int tokens, int D,
                                gar_stream_t stream) {
    GAR_CHECK_ARG(x && cls && pos && T_ > 0 && tokens > 0 && D > 0, "cls_pos_fill: bad args");
    dim3 grid((T_ * D + 255) / 256), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((cls_pos_fill_kernel<bf16_t>), grid, block, 0, s, (bf16_t*)x, (const bf16_t*)cls,
                           (const bf16_t*)pos, T_, tokens, D);
    else
        hipLaunchKernelGGL((cls_pos_fill_kernel<float>), grid, block, 0, s, (float*)x, (const float*)cls,
                           (const float*)pos, T_, tokens, D);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// x[t, token_offset + p, :] += add[t, p, :] — `x = x + mask_embeds.flatten(2).transpose(1, 2)` of the reference's
// custom_forward_features (modeling_perception_lm.py:195-196) for callers that hand the mask-embedding conv's output to
// mllm.get_image_features themselves (generate() never needs it: its mask conv is K columns of the patch-embed GEMM).
// One rounding in the tensor's dtype, like the reference's add.
template <typename T>
__global__ __launch_bounds__(256) void tokens_add_kernel(T* __restrict__ x, const T* __restrict__ add, int64_t n8, int tokens_in,
                                                         int tokens_out, int token_offset, int D) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const int64_t e = i * 8;                                  // element of add [T, tokens_in, D]
    const int d = (int)(e % D);
    const int64_t row = e / D;
    const int p = (int)(row % tokens_in);
    const int64_t t = row / tokens_in;
    T* xp = x + ((t * tokens_out + token_offset + p) * (int64_t)D + d);
    float a[8], b[8];
    ld8(xp, a);
    ld8(add + e, b);
#pragma unroll
    for (int k = 0; k < 8; ++k) a[k] += b[k];
    st8(xp, a);
}

extern "C" int gar_tokens_add(int dtype, void* x, const void* add, int T_, int tokens_in, int tokens_out, int token_offset, int D,
                              gar_stream_t stream) {
    GAR_CHECK_ARG(x && add && T_ > 0 && tokens_in > 0 && token_offset >= 0 && tokens_out >= tokens_in + token_offset && D > 0 &&
                      D % 8 == 0, "tokens_add: bad args");
    const int64_t n8 = (int64_t)T_ * tokens_in * D / 8;
    dim3 grid((unsigned)((n8 + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((tokens_add_kernel<bf16_t>), grid, block, 0, s, (bf16_t*)x, (const bf16_t*)add, n8, tokens_in, tokens_out,
                           token_offset, D);
    else
        hipLaunchKernelGGL((tokens_add_kernel<float>), grid, block, 0, s, (float*)x, (const float*)add, n8, tokens_in, tokens_out,
                           token_offset, D);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// vit_qkv_post: block = (tile t, head h, 64-token chunk); 256 threads = 32 tokens x 8 lanes x 8 elements, 2 passes.
// V goes through LDS so Vt rows (64 consecutive tokens of one d) are written as full lines.
// ---------------------------------------------------------------------------------------------------------------
template <typename T, int HD>
__global__ __launch_bounds__(256) void vit_qkv_post_kernel(const T* __restrict__ qkv, const float* __restrict__ sn,
                                                           const float* __restrict__ cs, T* __restrict__ Q,
                                                           T* __restrict__ K, T* __restrict__ Vt, int N, int npt, int H,
                                                           int Npad, float q_scale) {
    constexpr int LPT = HD / 8;                 // lanes per token
    __shared__ T vs[64 * (HD + 2)];
    const int chunks = Npad / 64;
    const int ch = blockIdx.x % chunks;
    const int h = (blockIdx.x / chunks) % H;
    const int t = blockIdx.x / (chunks * H);
    const int D = H * HD;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < 64 * LPT; idx += 256) {       // (token within chunk, 8-element group): any HD % 8 == 0
        const int nl = idx / LPT, d8 = (idx % LPT) * 8;
        const int n = ch * 64 + nl;
        float q[8], k[8], v[8];
        if (n < N) {
            const T* row = qkv + ((int64_t)t * N + n) * (3 * D) + h * HD + d8;
            ld8(row, q);
            ld8(row + D, k);
            ld8(row + 2 * D, v);
            if (n >= npt) {
                float s8[8], c8[8];
                const float* sp = sn + (int64_t)(n - npt) * HD + d8;
                const float* cp = cs + (int64_t)(n - npt) * HD + d8;
#pragma unroll
                for (int e = 0; e < 8; ++e) { s8[e] = sp[e]; c8[e] = cp[e]; }
                float qo[8], ko[8];
#pragma unroll
                for (int e = 0; e < 8; e += 2) {     // rot(x) = (-x[2i+1], x[2i])
                    qo[e] = q[e] * c8[e] + (-q[e + 1]) * s8[e];
                    qo[e + 1] = q[e + 1] * c8[e + 1] + q[e] * s8[e + 1];
                    ko[e] = k[e] * c8[e] + (-k[e + 1]) * s8[e];
                    ko[e + 1] = k[e + 1] * c8[e + 1] + k[e] * s8[e + 1];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) { q[e] = qo[e]; k[e] = ko[e]; }
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] *= q_scale;
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) q[e] = k[e] = v[e] = 0.f;
        }
        const int64_t o = (((int64_t)t * H + h) * Npad + n) * HD + d8;
        st8(Q + o, q);
        st8(K + o, k);
#pragma unroll
        for (int e = 0; e < 8; ++e) DT<T>::st(&vs[nl * (HD + 2) + d8 + e], v[e]);
    }
    __syncthreads();
    // Vt[t,h,d, ch*64 + 0..63]: thread (d = tid/8 + 32*i, 8 tokens)
    for (int d = tid / 8; d < HD; d += 32) {
        const int n8 = (tid % 8) * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = DT<T>::ld(&vs[(n8 + e) * (HD + 2) + d]);
        st8(Vt + (((int64_t)t * H + h) * HD + d) * Npad + ch * 64 + n8, o);
    }
}

extern "C" int gar_vit_qkv_post(int dtype, const void* qkv, const float* sn, const float* cs, void* Q, void* K, void* Vt,
                                int T_, int N, int npt, int H, int hd, int Npad, float q_scale, gar_stream_t stream) {
    GAR_CHECK_ARG(qkv && sn && cs && Q && K && Vt, "vit_qkv_post: null pointer");
    GAR_CHECK_ARG(Npad % 64 == 0 && Npad >= N && N > 0 && T_ > 0 && H > 0, "vit_qkv_post: bad shape");
    GAR_CHECK_ARG(hd == 64 || hd == 96 || hd == 128, "vit_qkv_post: head_dim %d not built (64, 96, 128)", hd);
    dim3 grid(T_ * H * (Npad / 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_VQP(TT, HD_)                                                                                          \
    hipLaunchKernelGGL((vit_qkv_post_kernel<TT, HD_>), grid, block, 0, s, (const TT*)qkv, sn, cs, (TT*)Q, (TT*)K,    \
                       (TT*)Vt, N, npt, H, Npad, q_scale)
    if (dtype == GAR_BF16) { if (hd == 64) LAUNCH_VQP(bf16_t, 64); else if (hd == 96) LAUNCH_VQP(bf16_t, 96); else LAUNCH_VQP(bf16_t, 128); }
    else { if (hd == 64) LAUNCH_VQP(float, 64); else if (hd == 96) LAUNCH_VQP(float, 96); else LAUNCH_VQP(float, 128); }
#undef LAUNCH_VQP
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// vit_v_transpose: the V third of vit_qkv_post for the fused qkv GEMM (GAR_EPI_QKV_ROPE writes q / k in place):
// V [T*N, H*HD] row-major -> Vt [T, H, HD, Npad] (zero for tokens >= N), through LDS so rows of Vt are full lines.
template <typename T, int HD>
__global__ __launch_bounds__(256) void vit_v_transpose_kernel(const T* __restrict__ V, T* __restrict__ Vt, int N, int H,
                                                              int Npad) {
    constexpr int LPT = HD / 8;
    __shared__ T vs[64 * (HD + 2)];
    const int chunks = Npad / 64;
    const int ch = blockIdx.x % chunks;
    const int h = (blockIdx.x / chunks) % H;
    const int t = blockIdx.x / (chunks * H);
    const int D = H * HD, tid = threadIdx.x;
    for (int idx = tid; idx < 64 * LPT; idx += 256) {
        const int nl = idx / LPT, d8 = (idx % LPT) * 8, n = ch * 64 + nl;
        float v[8];
        if (n < N) ld8(V + ((int64_t)t * N + n) * D + h * HD + d8, v);
        else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.f;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) DT<T>::st(&vs[nl * (HD + 2) + d8 + e], v[e]);
    }
    __syncthreads();
    for (int d = tid / 8; d < HD; d += 32) {
        const int n8 = (tid % 8) * 8;
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = DT<T>::ld(&vs[(n8 + e) * (HD + 2) + d]);
        st8(Vt + (((int64_t)t * H + h) * HD + d) * Npad + ch * 64 + n8, o);
    }
}

extern "C" int gar_vit_v_transpose(int dtype, const void* V, void* Vt, int T_, int N, int H, int hd, int Npad,
                                   gar_stream_t stream) {
    GAR_CHECK_ARG(V && Vt, "vit_v_transpose: null pointer");
    GAR_CHECK_ARG(Npad % 64 == 0 && Npad >= N && N > 0 && T_ > 0 && H > 0, "vit_v_transpose: bad shape");
    GAR_CHECK_ARG(hd == 64 || hd == 96 || hd == 128, "vit_v_transpose: head_dim %d not built (64, 96, 128)", hd);
    dim3 grid(T_ * H * (Npad / 64)), block(256);
    hipStream_t s = (hipStream_t)stream;
#define LAUNCH_VT(TT, HD_) \
    hipLaunchKernelGGL((vit_v_transpose_kernel<TT, HD_>), grid, block, 0, s, (const TT*)V, (TT*)Vt, N, H, Npad)
    if (dtype == GAR_BF16) { if (hd == 64) LAUNCH_VT(bf16_t, 64); else if (hd == 96) LAUNCH_VT(bf16_t, 96); else LAUNCH_VT(bf16_t, 128); }
    else { if (hd == 64) LAUNCH_VT(float, 64); else if (hd == 96) LAUNCH_VT(float, 96); else LAUNCH_VT(float, 128); }
#undef LAUNCH_VT
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pool2x2: one thread per (out token, 8 channels); reads 4 x 16 B, writes 16 B.
// algorithmic bytes = T*g*g*C*sizeof(T) read + a quarter of that written.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void pool2x2_kernel(const T* __restrict__ x, T* __restrict__ y, int T_, int g, int C,
                                                      int in_tile_tokens, int in_token_offset) {
    const int c8n = C / 8;
    const int go = g / 2;
    const int64_t total = (int64_t)T_ * go * go * c8n;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c8 = (int)(i % c8n);
    const int64_t tok = i / c8n;
    const int ox = (int)(tok % go), oy = (int)((tok / go) % go), t = (int)(tok / ((int64_t)go * go));
    const T* base = x + ((int64_t)t * in_tile_tokens + in_token_offset) * C + c8 * 8;
    float a[8], b[8], c[8], d[8], o[8];
    ld8(base + (int64_t)((2 * oy) * g + 2 * ox) * C, a);
    ld8(base + (int64_t)((2 * oy) * g + 2 * ox + 1) * C, b);
    ld8(base + (int64_t)((2 * oy + 1) * g + 2 * ox) * C, c);
    ld8(base + (int64_t)((2 * oy + 1) * g + 2 * ox + 1) * C, d);
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (((a[e] + b[e]) + c[e]) + d[e]) * 0.25f;
    st8(y + tok * C + c8 * 8, o);
}

extern "C" int gar_pool2x2(int dtype, const void* x, void* y, int T_, int g, int C, int in_tile_tokens,
                           int in_token_offset, gar_stream_t stream) {
    GAR_CHECK_ARG(x && y && T_ > 0 && g > 0 && g % 2 == 0 && C % 8 == 0, "pool2x2: bad args");
    if (in_tile_tokens <= 0) in_tile_tokens = g * g;
    GAR_CHECK_ARG(in_token_offset >= 0 && in_token_offset + g * g <= in_tile_tokens, "pool2x2: bad token window");
    const int64_t total = (int64_t)T_ * (g / 2) * (g / 2) * (C / 8);
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == GAR_BF16)
        hipLaunchKernelGGL((pool2x2_kernel<bf16_t>), grid, block, 0, s, (const bf16_t*)x, (bf16_t*)y, T_, g, C,
                           in_tile_tokens, in_token_offset);
    else
        hipLaunchKernelGGL((pool2x2_kernel<float>), grid, block, 0, s, (const float*)x, (float*)y, T_, g, C,
                           in_tile_tokens, in_token_offset);
    GAR_CHECK_LAUNCH();
    return GAR_OK;
}
