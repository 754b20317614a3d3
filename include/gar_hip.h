/*
 * gar_hip.h — C ABI of libgar_hip.so: the MI355X (gfx950 / CDNA4) device side of the Grasp-Any-Region
 * region-captioning hot path.
 *
 * The reference (Haochen-Wang409/Grasp-Any-Region) is pure Python; it has no FFI of its own. Each entry
 * point below therefore replaces a *third-party GPU op call site* on the reference's hot path, cited as
 * file:line under /root/reference (see INTEGRATION.md for the binding a maintainer adds on the reference side).
 *
 * Conventions
 *   - plain C: raw device pointers + explicit shapes; no torch types; no allocation inside; no global state
 *     other than a thread-local last-error string.
 *   - every function returns 0 (GAR_OK) or a negative error code and never throws across the ABI.
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and nothing synchronises
 *     (safe for hipGraph capture).
 *   - `dtype` selects the storage type of activations/weights: GAR_F32 (parity mode, exact-f32 MFMA/VALU math)
 *     or GAR_BF16 (performance mode, bf16 storage, fp32 accumulation). Index tensors are int64 or int32 as stated.
 *     fp16 (the reference's `--data_type fp16`, demo/gar_with_mask.py:41-45) is the TWIN library libgar_hip_f16.so: the same
 *     sources compiled with -DGAR_HALF_F16=1 (csrc/common.h), the same entry points and layouts, in which dtype code 1
 *     means IEEE binary16 storage (conversions round to nearest even, the matrix instructions are the _f16 forms, the
 *     attention's stored softmax weights stay below 2^15). A host binds one handle per element type (gar_amd/hip.py lib()).
 *   - matrices are row-major; nn.Linear weights keep the PyTorch [out_features, in_features] layout.
 */
#ifndef GAR_HIP_H
#define GAR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GAR_OK 0
#define GAR_ERR_ARG (-1)
#define GAR_ERR_LAUNCH (-2)
#define GAR_ERR_ARCH (-3)
#define GAR_ERR_UNSUPPORTED (-4)

#define GAR_F32 0
#define GAR_BF16 1

#define GAR_ABI_VERSION 15

/* GEMM epilogues */
#define GAR_EPI_NONE 0            /* C = A W^T                                               */
#define GAR_EPI_BIAS 1            /* C = A W^T + bias                                        */
#define GAR_EPI_BIAS_GELU 2       /* C = gelu_erf(A W^T + bias)                              */
#define GAR_EPI_BIAS_SCALE_RES 3  /* C = residual + gamma * (A W^T + bias)   (LayerScale)    */
#define GAR_EPI_RES 4             /* C = residual + A W^T                                    */
#define GAR_EPI_SWIGLU 5          /* W rows interleaved in blocks of 16: [gate16|up16]...;   */
                                  /* C[:, N/2] = silu(gate) * up                             */
#define GAR_EPI_PATCH_POS 6       /* row m -> tile m/tin, token tok_off + m%tin of a          */
                                  /* [tiles, tout, N] output; C = A W^T + pos[tok_off+m%tin] */
#define GAR_EPI_QKV_ROPE 7        /* timm AttentionRope front half fused into the qkv GEMM (bf16 tile GEMM only):  */
                                  /* v = A W^T + bias, columns [q | k | v] x [head][head_dim]; row m = token        */
                                  /* m % qkv_tokens of image tile m / qkv_tokens. q, k: interleaved-pair RoPE for   */
                                  /* tokens >= qkv_prefix (tables [tokens - prefix, head_dim] f32), q * qkv_q_scale, */
                                  /* written to qkv_q / qkv_k [tiles, heads, qkv_tokens_pad, head_dim]; v written   */
                                  /* row-major to C [M, heads*head_dim] (transposed later by gar_vit_v_transpose),  */
                                  /* or head-major to qkv_v when that is set (gar_attention_vrow reads it in place)  */
#define GAR_EPI_QKV_ROPE_LLM 8    /* HF Llama pre-attention step fused into the qkv GEMM (bf16 tile GEMM only; ABI v9): what   */
                                  /* gar_llm_qkv_post does to the GEMM's output, done on the fp32 accumulators instead — no  */
                                  /* [M, (Hq+2Hkv) hd] intermediate. Columns [q heads | k heads | v heads] x head_dim (64 /  */
                                  /* 128), row m = token m % qkv_tokens of sequence m / qkv_tokens. q, k: half-split RoPE    */
                                  /* (x[d], x[d + hd/2]) at position pos0 + token - qkv_left_pad[seq] (tables qkv_cos /      */
                                  /* qkv_sin [max_pos, hd/2] f32), q * qkv_q_scale -> qkv_q [B, qkv_heads, qkv_tokens_pad,   */
                                  /* hd]; k, v -> rows pos0 + token of the caches qkv_k / qkv_v [B, qkv_kv_heads,            */
                                  /* qkv_kv_stride, hd]. A wave's 64-column strip must hold a rotation's both halves:        */
                                  /* within every q / k head the W rows are ordered so that strip j (64 j .. 64 j + 63)      */
                                  /* holds dims [32 j, 32 j + 32) then [hd/2 + 32 j, hd/2 + 32 j + 32) — the natural order   */
                                  /* for head_dim 64, [0..31, 64..95, 32..63, 96..127] for head_dim 128; v heads natural.    */
                                  /* No bias. GAR_ERR_UNSUPPORTED (nothing launched) when the tile GEMM does not take the    */
                                  /* problem: the caller then runs GAR_EPI_NONE + gar_llm_qkv_post (natural W order).        */

typedef void* gar_stream_t;

int gar_abi_version(void);
const char* gar_last_error(void);
/* 0 if `device` is a gfx950 part, GAR_ERR_ARCH otherwise (the library holds gfx950 code objects only). */
int gar_check_device(int device);

typedef struct gar_gemm_params {
    const void* A;  int64_t lda;       /* [M, K]                                                   */
    const void* W;  int64_t ldw;       /* [N, K]  (nn.Linear weight)                               */
    void* C;        int64_t ldc;       /* [M, N]  ([M, N/2] for SWIGLU)                            */
    int32_t M, N, K;                   /* K % 64 == 0 (bf16) / K % 16 == 0 (f32); N % 16 == 0      */
    int32_t epilogue;                  /* GAR_EPI_*                                                */
    const void* bias;                  /* [N] or NULL                                              */
    const void* residual; int64_t ldr; /* [M, N]                                                   */
    const void* gamma;                 /* [N]                                                      */
    const void* pos;                   /* PATCH_POS: [tokens_out, N]                               */
    int32_t tokens_in, tokens_out, token_offset;
    float norm_eps;                    /* with norm_w: rsqrt(mean(x^2) + norm_eps)                  */
    const void* norm_w;                /* [K] or NULL; M <= 16 only: C = epilogue(RMSNorm(A; norm_w) W^T), i.e. the */
                                       /* HF LlamaRMSNorm in front of q/k/v, gate/up and lm_head fused into the GEMM */
    /* GAR_EPI_QKV_ROPE only */
    void* qkv_q; void* qkv_k;          /* [tiles, heads, qkv_tokens_pad, head_dim]                                   */
    const float* qkv_sin; const float* qkv_cos;   /* [qkv_tokens - qkv_prefix, head_dim]; or qkv_cos == NULL and qkv_sin =   */
                                                  /* compact (sin, cos) pairs [1 + qkv_tokens - qkv_prefix, head_dim/2][2]   */
                                                  /* when the tables repeat each value for both elements of a rotated pair   */
                                                  /* (timm); ABI 14: row 0 is the identity (0, 1) — what the prefix rows and */
                                                  /* the v columns read — and token t's angles are row t - qkv_prefix + 1    */
    int32_t qkv_heads, qkv_head_dim, qkv_tokens, qkv_tokens_pad, qkv_prefix;
    float qkv_q_scale;
    /* Split-K for the decode GEMMs (bf16, M <= 64, GAR_EPI_NONE, no norm_w): with split_k > 1 the K range is cut into
     * split_k equal slices (K % (64 split_k) == 0, N % 4 == 0), every slice's fp32 product goes to `partial`
     * [split_k][M][N] (C is not written) and gar_splitk_residual_rmsnorm — the launch that follows anyway — sums them.
     * No atomics and no fences: a narrow output (Llama `down`: 128 weight tiles) then streams from 4x the workgroups. */
    int32_t split_k; float* partial;
    /* GAR_EPI_QKV_ROPE: when not NULL, v goes here head-major [tiles, heads, qkv_tokens_pad, head_dim] — the layout of
     * qkv_k, consumed by gar_attention_vrow — instead of row-major to C (which then is not written) */
    void* qkv_v;
    /* A LayerNorm / RMSNorm folded into the GEMM pair around it (bf16 tile GEMM; ABI v8). The normalisation of the residual
     * stream x is linear up to a per-row scale: LN(x) W^T + b = rstd_m * (x W''^T) + b' with W'' = (W diag(gamma)) minus each
     * row's mean (the mean subtraction is absorbed by centring the weight rows), b' = b + W beta; RMSNorm: W'' = W diag(g).
     *   row_scale [M] fp32: C = epilogue(row_scale[m] * (A W^T) + bias) — the CONSUMER (qkv / fc1 / gate-up GEMM reading x
     *     itself; GAR_EPI_NONE, BIAS, BIAS_GELU, SWIGLU, QKV_ROPE with the compact table, QKV_ROPE_LLM);
     *   row_stats [M][ceil(N/64)][2] fp32: per row and 64-column strip (sum y, sum y^2) of the bf16-rounded outputs y — the
     *     PRODUCER of x (GAR_EPI_RES, GAR_EPI_BIAS_SCALE_RES); gar_row_stats_finalize turns them into row_scale. */
    const float* row_scale;
    float* row_stats;
    /* GAR_EPI_QKV_ROPE_LLM only (ABI v9); it also uses qkv_q / qkv_k / qkv_v / qkv_sin / qkv_cos / qkv_heads (= Hq) /
     * qkv_head_dim / qkv_tokens (= S) / qkv_tokens_pad (= Spad of Q) / qkv_q_scale with the meanings given at the define.
     * qkv_pos_dev (device int32[1]) overrides qkv_pos0 when not NULL; qkv_left_pad device int32 [B] or NULL. */
    int32_t qkv_kv_heads, qkv_kv_stride, qkv_pos0;
    const int32_t* qkv_pos_dev;
    const int32_t* qkv_left_pad;
    /* bf16, M <= 64 (decode GEMMs; GAR_EPI_NONE / BIAS / SWIGLU), ABI v9: the RMSNorm in front of the GEMM with its gain folded
     * into W (W = W0 diag(g), the weight row_scale's consumers use): C = epilogue(rsqrt(mean_k A[m][k]^2 + norm_eps) * (A W^T)).
     * The row sums of squares are taken from the A tiles the kernel streams anyway (one extra MFMA pair per tile): no norm
     * launch, no normalised copy of A. */
    int32_t norm_folded;
} gar_gemm_params;

/* Replaces: every nn.Linear / cuBLAS GEMM on the path — timm Eva qkv/proj/fc1/fc2 (via
 * modeling_perception_lm.py:210-214), projector linear_1/2 (modeling_perception_lm.py:85-92), Llama
 * q/k/v/o/gate/up/down + lm_head (modeling_gar.py:418-426 -> HF LlamaModel), and the two 14x14/stride-14
 * convolutions as one im2col GEMM (modeling_gar.py:326-328, modeling_perception_lm.py:194-196). */
int gar_gemm(int dtype, const gar_gemm_params* p, gar_stream_t stream);
/* 1 when gar_gemm runs `p` on the persistent 256 x 256 bf16 tile GEMM — the only kernel with the folded-norm epilogues
 * (row_scale / row_stats) and the fused qkv forms — 0 otherwise; nothing is launched (ABI v10). The host plans a pass around
 * those epilogues only after asking this predicate, so the conditions (>= 128 output tiles, N >= 256, 16-byte alignment,
 * ldc / ldr % 8, operands < 4 GiB, epilogue / row_scale pairing) exist once, in the library. */
int gar_gemm_tile_takes(int dtype, const gar_gemm_params* p);

/* A1+A2+K2 input side: mask decode round((m+1)/2*255)->clamp[0,P]->(v!=P) (modeling_gar.py:315-327) and patch
 * extraction of both `pixel_values` and the binary mask into one GEMM operand:
 *   out[T*g*g, Kp]: cols [0,3pp) = pixel patch in (c,ky,kx) order, [3pp,6pp) = binary mask patch, rest 0.
 * `mask` may be NULL (then the mask columns are 0). */
int gar_patch_im2col(int dtype, const void* pixel, const void* mask, void* out, int T, int img, int patch,
                     int Kp, int prompt_numbers, gar_stream_t stream);

/* A1 alone: out = (clamp(round((m+1)/2*255), 0, P) != P) ? 1 : 0 in the tensor's dtype, elementwise over n elements
 * (n % 8 == 0) — the mask decode of modeling_gar.py:315-327 for gar_patch_embed, which reads the images in place. */
int gar_mask_decode(int dtype, const void* mask, void* out, int64_t n, int prompt_numbers, gar_stream_t stream);

/* A2 + K2 with the patch tiles DMA'd from the image tiles straight into LDS (bf16 tile GEMM; no im2col matrix in HBM):
 *   x[t, token_offset + (py*g + px), :] = conv14(pixel)[t, :, py, px] + conv14(maskbin)[t, :, py, px] + pos[token_offset + py*g + px]
 * i.e. `mask_patch_embedding` (modeling_gar.py:54-60,326-328) + timm PatchEmbed + `x + mask_embeds` + pos_embed
 * (modeling_perception_lm.py:194-197) as ONE GEMM whose A operand is gathered from `pixel` / `maskbin`
 * [T, 3, img, img] (maskbin = gar_mask_decode output). Wg [D, gar_patch_embed_k(img, patch)] holds both conv weights in the
 * gather's K order: column ((tensor*3 + c)*4 + ky/4)*64 + (ky%4)*16 + kx  <-  W_tensor[d, c, ky, kx], zero elsewhere
 * (tensor 0 = patch_embed.proj, 1 = mask_patch_embedding). x rows have pitch D, tiles have tokens_out rows.
 * Returns GAR_ERR_UNSUPPORTED (nothing launched) unless bf16, img / patch == 32, patch even and <= 16, and the
 * problem has >= 128 output tiles of 256 x 256; the caller then uses gar_patch_im2col + gar_gemm(GAR_EPI_PATCH_POS). */
int gar_patch_embed_k(int img, int patch);
int gar_patch_embed(int dtype, const void* pixel, const void* maskbin, const void* Wg, const void* pos, void* x, int T,
                    int img, int patch, int D, int tokens_out, int token_offset, gar_stream_t stream);

/* cls token row: x[t, 0, :] = cls + pos[0]   (timm Eva._pos_embed, via modeling_perception_lm.py:197) */
int gar_cls_pos_fill(int dtype, void* x, const void* cls, const void* pos, int T, int tokens, int D,
                     gar_stream_t stream);
/* x[t, token_offset + p, :] += add[t, p, :] for x [T, tokens_out, D], add [T, tokens_in, D] (D % 8 == 0): the reference's
 * `x = x + mask_embeds.flatten(2).transpose(1, 2)` (modeling_perception_lm.py:195-196) for callers of
 * mllm.get_image_features(pixel_values, mask_embeds=...) (modeling_gar.py:334-337) that computed the mask-embedding conv
 * themselves; generate() folds that conv into the patch-embed GEMM instead (ABI v10). */
int gar_tokens_add(int dtype, void* x, const void* add, int T, int tokens_in, int tokens_out, int token_offset, int D,
                   gar_stream_t stream);

/* nn.LayerNorm (norm_pre / norm1 / norm2 of the PE ViT) and HF LlamaRMSNorm. y may alias x.
 * ldx / ldy = row strides in elements (<= 0 means D). */
int gar_layernorm(int dtype, const void* x, void* y, const void* w, const void* b, int M, int D, int64_t ldx,
                  int64_t ldy, float eps, gar_stream_t stream);
/* The statistics half of a norm whose scaling half is folded into the next GEMM (gar_gemm_params.row_scale):
 * gar_row_rstd: rstd[m] = rsqrt(var(x[m, :]) + eps) (rms = 0; two-pass variance) or rsqrt(mean(x^2) + eps) (rms = 1) straight
 *   from x — the first norm of a chain, whose input no GEMM epilogue produced;
 * gar_row_stats_finalize: the same from the (sum, sum of squares) partials a producer GEMM wrote (row_stats, `strips`
 *   pairs per row, summed in strip order: deterministic). timm LayerNorm / HF LlamaRMSNorm statistics, fp32. */
int gar_row_rstd(int dtype, const void* x, int M, int D, int64_t ldx, float eps, int rms, float* rstd, gar_stream_t stream);
int gar_row_stats_finalize(const float* row_stats, int M, int strips, int D, float eps, int rms, float* rstd,
                           gar_stream_t stream);
int gar_rmsnorm(int dtype, const void* x, void* y, const void* w, int M, int D, int64_t ldx, int64_t ldy, float eps,
                gar_stream_t stream);
/* The reduction of a split-K decode GEMM fused with what follows it in a Llama layer (HF LlamaDecoderLayer:
 * hidden = residual + mlp(...); next layer's input_layernorm — modeling_gar.py:418-426 -> transformers LlamaModel):
 *   h[m, :] = round(h[m, :] + sum_s partial[s][m][:])   (slices added in order s = 0, 1, ...; one rounding, like GAR_EPI_RES)
 *   y[m, :] = LlamaRMSNorm(h[m, :]; w)                  (y == NULL: residual update only)
 * bf16, M <= 64 rows, D % 8 == 0, D <= 4096, split_k <= 8 (<= 4 when D > 2048); h and y contiguous [M, D]. */
int gar_splitk_residual_rmsnorm(int dtype, const float* partial, int split_k, void* h, const void* w, void* y, int M,
                                int D, float eps, gar_stream_t stream);

/* timm AttentionRope pre-attention step on the fused qkv output [T*N, 3*H*hd]: interleaved-pair 2-D RoPE on q,k
 * for tokens >= npt (tables sin/cos [N-npt, hd] f32), softmax scale*log2(e) folded into q, and re-layout to
 *   Q [T,H,Npad,hd], K [T,H,Npad,hd], Vt [T,H,hd,Npad]   (Vt pad columns zeroed). */
int gar_vit_qkv_post(int dtype, const void* qkv, const float* sin, const float* cos, void* Q, void* K, void* Vt,
                     int T, int N, int npt, int H, int hd, int Npad, float q_scale, gar_stream_t stream);

/* V third of gar_vit_qkv_post, for the fused GAR_EPI_QKV_ROPE GEMM: V [T*N, H*hd] row-major -> Vt [T, H, hd, Npad]
 * (zero for tokens >= N). */
int gar_vit_v_transpose(int dtype, const void* V, void* Vt, int T, int N, int H, int hd, int Npad, gar_stream_t stream);

/* HF Llama pre-attention step on fused qkv [B*S, (Hq+2Hkv)*hd]: half-split RoPE (cos/sin [max_pos, hd/2] f32,
 * row = absolute position pos0+s), q scale folded, Q [B,Hq,Spad,hd]; K and V appended to the cache at
 * positions pos0..pos0+S-1:  Kc and Vc [B,Hkv,Smax,hd] both (ABI v9; through v8 V was kept transposed): an appended token
 * is one contiguous row of either cache, and the attention kernels transpose V on their LDS reads.
 * If `pos_dev` != NULL the start position is read from device memory (pos_dev[0]) instead of pos0 (graph replay).
 * `left_pad` (device int32 [B] or NULL): a LEFT-PADDED batch as HF generation builds it from `attention_mask`
 * (modeling_gar.py:418-426 forwards it): sequence b's first real token sits at row left_pad[b]; a token keeps its row in
 * the cache and rotates by position (row - left_pad[b]) (HF: position_ids = cumsum(attention_mask) - 1).
 * `qk_strip_order` != 0 (ABI v10): the q / k head columns of `qkv` come in GAR_EPI_QKV_ROPE_LLM's strip order (head_dim 128:
 * dims [0..31, 64..95, 32..63, 96..127]; natural for head_dim 64) — what a GEMM with the strip-ordered (folded) qkv weight,
 * the only bf16 copy the host keeps, produces when the fused epilogue does not take the shape. v heads are natural. */
int gar_llm_qkv_post(int dtype, const void* qkv, const float* cos, const float* sin, void* Q, void* Kc, void* Vc,
                     int B, int S, int Spad, int Hq, int Hkv, int hd, int Smax, int pos0, const int32_t* pos_dev,
                     const int32_t* left_pad, float q_scale, int qk_strip_order, gar_stream_t stream);

/* Flash-style attention (replaces F.scaled_dot_product_attention in timm AttentionRope and flash-attn-2 /
 * eager attention in HF Llama, modeling_gar.py:40-43). Q [B,Hq,q_pad,hd] (pre-scaled by scale*log2e),
 * K [B,Hkv,kv_stride,hd], Vt [B,Hkv,hd,kv_stride] (entries beyond kv_len must be finite); O [B*q_len, Hq*hd]
 * token-major. causal: query i attends kv j <= i + (kv_len - q_len).
 * If `kv_len_dev` != NULL the kv length is read from device memory (decode step inside a replayed graph).
 * `kv_start` (device int32 [B] or NULL = 0): first visible kv row of sequence b — the padding keys of a left-padded
 * batch stay hidden (HF combines the causal mask with `attention_mask`): query i sees kv_start[b] <= j <= max(i + kv_len -
 * q_len, kv_start[b]); rows in front of kv_start[b] are padding queries (finite output nobody reads). */
int gar_attention(int dtype, const void* Q, const void* K, const void* Vt, void* O, int B, int Hq, int Hkv, int hd,
                  int q_len, int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev,
                  const int32_t* kv_start, gar_stream_t stream);
/* The same with V row-major, [B,Hkv,kv_stride,hd] like K (what GAR_EPI_QKV_ROPE writes to gar_gemm_params.qkv_v): the PV
 * operand is formed by gfx950's transposing LDS read (ds_read_b64_tr_b16), so timm AttentionRope's v needs no transpose
 * pass between the qkv GEMM and the attention (modeling_perception_lm.py:210-214 -> timm Eva attention), and the Llama KV
 * cache (gar_llm_qkv_post) keeps V in K's layout. bf16: head_dim 64 / 96 / 128; f32 (parity mode): head_dim 64 / 128, the
 * tile is transposed while it is staged; GAR_ERR_UNSUPPORTED (nothing launched) otherwise.
 * kv_prefix = 1 (non-causal only): key / value row 0 — the ViT's cls token — enters through the initial softmax state
 * (m0 = q.k0, l0 = 1, O0 = v0) and the kv tiles cover rows 1 .. kv_len - 1: 1 + 1024 keys are 16 tiles, not 17. Same
 * result up to the fp32 summation order. */
int gar_attention_vrow(int dtype, const void* Q, const void* K, const void* V, void* O, int B, int Hq, int Hkv, int hd,
                       int q_len, int q_pad, int kv_len, int kv_stride, int causal, const int32_t* kv_len_dev,
                       const int32_t* kv_start, int kv_prefix, gar_stream_t stream);

/* Single-token decode attention over the KV cache (the per-token LlamaModel step of HF's greedy loop,
 * modeling_gar.py:418-426): split-KV, the Hq/Hkv query heads of a kv head share one pass over K / V.
 * q: the query row of (b, head) starts at q + (b*Hq + head) * q_stride elements (q_stride = hd, or 0, for the packed
 * [B,Hq,hd] form of a decode step; q_stride = Spad*hd with q = &Q[0][0][S-1][0] reads the LAST prompt row of a prefill's
 * Q [B,Hq,Spad,hd] in place — the pruned last prefill layer, ABI v10), pre-scaled; Kc, Vc [B,Hkv,Smax,hd]
 * (gar_llm_qkv_post's caches); kv length = kv_len_dev[0] (device memory, so
 * one captured hipGraph replays for every token); O [B, Hq*hd]. workspace >= gar_attention_decode_workspace() bytes.
 * GAR_F32 routes to gar_attention_vrow. bf16: head_dim 64 / 128, Smax * hd * 2 < 2 GiB (GAR_ERR_UNSUPPORTED otherwise).
 * `kv_start` (device int32 [B] or NULL): sequence b's keys are cache rows kv_start[b] .. kv_len - 1 (left-padded batch);
 * the kv splits divide that range. */
int64_t gar_attention_decode_workspace(int B, int Hq, int hd, int max_splits);
int gar_attention_decode(int dtype, const void* q, int64_t q_stride, const void* Kc, const void* Vc, void* O, int B, int Hq,
                         int Hkv, int hd, int Smax, const int32_t* kv_len_dev, const int32_t* kv_start, int max_splits,
                         void* workspace, gar_stream_t stream);
/* gar_llm_qkv_post (S = 1) and gar_attention_decode as ONE launch (ABI v9; bf16): `qkv` [B, (Hq+2Hkv)*hd] is the step's raw
 * qkv GEMM output. Every workgroup applies the half-split RoPE at position pos_dev[0] - left_pad[b] and q_scale to its query
 * heads while it loads them; the workgroup that owns the last kv tile of (b, kv head) rotates the new key, appends key and
 * value to cache row pos_dev[0] of Kc / Vc and attends over rows left_pad[b] .. pos_dev[0]. Same arithmetic, bit for bit,
 * as the two calls (HF: apply_rotary_pos_emb + DynamicCache.update + attention of one greedy step, modeling_gar.py:418-426).
 * GAR_ERR_UNSUPPORTED (nothing launched) for GAR_F32: parity mode keeps the two calls.
 * `qk_strip_order` as in gar_llm_qkv_post (ABI v10): the decode step's qkv GEMM runs on the strip-ordered folded weight. */
int gar_attention_decode_qkv(int dtype, const void* qkv, const float* cos, const float* sin, void* Kc, void* Vc, void* O,
                             int B, int Hq, int Hkv, int hd, int Smax, const int32_t* pos_dev, const int32_t* left_pad,
                             float q_scale, int qk_strip_order, int max_splits, void* workspace, gar_stream_t stream);

/* PerceptionLMAdaptiveAvgPooling (modeling_perception_lm.py:47-60): per tile [g*g, C] -> [(g/2)^2, C], exact 2x2
 * mean. Input tile t starts at row t*in_tile_tokens + in_token_offset of x (lets the projector run over the
 * cls-inclusive token matrix and drop the cls row here, modeling_perception_lm.py:263-264). */
int gar_pool2x2(int dtype, const void* x, void* y, int T, int g, int C, int in_tile_tokens, int in_token_offset,
                gar_stream_t stream);

/* get_placeholder_mask + crop-token span search (modeling_perception_lm.py:271-331, modeling_gar.py:356-360),
 * sync-free: slot[s] = rank of s among image tokens of its row or -1; counts[b] = #image tokens;
 * spans[b][c] = (min index, max index) of crop token c or (-1,-1). input_ids int64 [B,S]. */
int gar_placeholder_scan(const int64_t* input_ids, int B, int S, int64_t image_token_id, const int64_t* crop_ids,
                         int n_crop, int32_t* slot, int32_t* counts, int32_t* spans, int32_t* rank_pos, int rank_stride,
                         gar_stream_t stream);
/* rank_pos (nullable) [B, rank_stride]: the inverse of slot — position of the image token of rank r (ranks >= rank_stride
 * are dropped); what gar_roi_replay_inplace reads pooled features through. */

/* PerceptionLMAdaptiveAvgPooling + nn.Embedding + masked_scatter in one pass (modeling_perception_lm.py:47-60,
 * modeling_gar.py:332,341-346): out[b,s,:] = slot[b,s] < 0 ? E[id] : 2x2 mean of the projector output `proj`
 * [B * tiles_per_sample * in_tile_tokens, C] at pooled token slot[b,s] (tile = slot / (g/2)^2; the g x g grid of a tile
 * starts at its row in_token_offset, which drops a cls row as gar_pool2x2 does). The pooled features are not
 * materialised; gar_roi_replay_inplace reads them back from `out`. */
int gar_pool_assemble(int dtype, const int64_t* input_ids, const int32_t* slot, const void* E, const void* proj, void* out,
                      int B, int S, int C, int tiles_per_sample, int g, int in_tile_tokens, int in_token_offset,
                      int64_t vocab, gar_stream_t stream);

/* nn.Embedding + masked_scatter (modeling_gar.py:332,341-346): out[b,s,:] = slot<0 ? E[id] : feats[b][slot].
 * feats [B, n_feat_rows, C] */
int gar_embed_assemble(int dtype, const int64_t* input_ids, const int32_t* slot, const void* E, const void* feats,
                       void* out, int B, int S, int C, int64_t n_feat_rows, int64_t vocab, gar_stream_t stream);

/* RoI-aligned feature replay (modeling_gar.py:348-414 == _merge + torchvision.ops.roi_align(aligned=True,
 * sampling_ratio=2) + permute/flatten/cast + splice), reading the pooled features in their TILE layout
 * feats [tiles(incl. thumbnail), P*P, C] (no _merge copy, no fp32 copy of the map) and writing P*P rows of
 * `embeds` [S, C] starting at spans[2*crop_index] (device) — skipped when the crop token is absent.
 * roi = (x1,y1,x2,y2) in the reference's `roi_feat` coordinates, spatial_scale as passed to roi_align. */
int gar_roi_replay(int dtype, const void* feats, void* embeds, const int32_t* spans, int crop_index, int first_tile,
                   int ncw, int nch, int P, int C, int S, float rx1, float ry1, float rx2, float ry2,
                   float spatial_scale, int sampling_ratio, int aligned, gar_stream_t stream);

/* The same replay for every crop token of every sample of a batch in ONE launch (the reference loops
 * `for batch_idx` / `for crop_token` in Python, modeling_gar.py:348-356). feats [B, tiles_per_sample, P*P, C],
 * embeds [B, S, C], spans [B, n_crop, 2] (from gar_placeholder_scan); jobs: device array, one entry per
 * (sample, crop token that has a bbox). */
typedef struct gar_roi_job {
    int32_t sample;       /* row of the batch */
    int32_t crop_index;   /* index into the crop-token list given to gar_placeholder_scan */
    int32_t first_tile;   /* first tile of the merged map inside the sample (1 = skip thumbnail; video: frame) */
    int32_t ncw, nch;     /* tiles per row / column of the merged map */
    float x1, y1, x2, y2; /* roi in `roi_feat` coordinates (modeling_gar.py:366-387) */
    float spatial_scale;  /* as passed to roi_align (:393) */
} gar_roi_job;
int gar_roi_replay_batched(int dtype, const void* feats, void* embeds, const int32_t* spans, const gar_roi_job* jobs,
                           int n_jobs, int n_crop, int tiles_per_sample, int P, int C, int S, int sampling_ratio,
                           int aligned, gar_stream_t stream);

/* The batched replay reading the pooled features IN PLACE: after gar_pool_assemble the pooled token of rank r of sample b
 * is row rank_pos[b][r] of embeds[b] (rank = tile * P*P + token; rank_pos from gar_placeholder_scan), so the merged map
 * of modeling_gar.py:350-352 is addressed inside the sequence itself and no [tiles, P*P, C] feature tensor exists. The
 * rows it writes (crop-token spans) are disjoint from the rows it reads (image-token rows). */
int gar_roi_replay_inplace(int dtype, void* embeds, const int32_t* spans, const int32_t* rank_pos, int rank_stride,
                           const gar_roi_job* jobs, int n_jobs, int n_crop, int P, int C, int S, int sampling_ratio,
                           int aligned, gar_stream_t stream);

/* GPU-side preprocessing (PerceptionLMImageProcessorFast.resize -> _split -> rescale_and_normalize,
 * image_processing_perception_lm_fast.py:268-372; NEAREST for the visual-prompt id matrix, eval_dataset.py:122-139).
 * src: uint8 RGB image [H, W, 3] on the device. Separable antialiased bicubic: pass 1 (horizontal) writes fp32
 * tmp [3, H, Wout]; pass 2 (vertical) rounds half-to-even, clamps to [0,255], normalises (v - 255 mean) / (255 std) (HF fused rescale_and_normalize) and
 * writes tiles out[tile0 + (yo/ts)*ncw + (xo/ts)][c][yo%ts][xo%ts] in `dtype`. Tap tables (first source index,
 * tap count, weights [n_out, kmax]) come from the host. Accumulation order: t = s[0]*w[0]; t = fma(s[j], w[j], t). */
int gar_resize_bicubic_h(const uint8_t* src, float* tmp, int H, int W, int Wout, const int32_t* xmin,
                         const int32_t* xsize, const float* wx, int kmax, gar_stream_t stream);
int gar_resize_bicubic_v_tiles(int dtype, const float* tmp, void* out, int H, int Wout, int Hout, int ts, int ncw,
                               int tile0, const int32_t* ymin, const int32_t* ysize, const float* wy, int kmax,
                               float mean, float stdv, gar_stream_t stream);
/* nearest: out pixel (yo, xo) = src[yi[yo]][xi[xo]] (index tables from the host), same normalisation and tiling */
int gar_resize_nearest_tiles(int dtype, const uint8_t* src, void* out, int H, int W, int Hout, int Wout, int ts,
                             int ncw, int tile0, const int32_t* xi, const int32_t* yi, float mean, float stdv,
                             gar_stream_t stream);

/* COCO compressed run-length mask -> row-major [h, w] bytes (host memory, CPU): what `pycocotools.mask.decode`
 * does for the `mask_rles` / `segmentation` entries the benchmark loops read (evaluation/GAR-Bench/inference.py:142-145,
 * evaluation/DLC-Bench/inference.py:121-122). Returns the foreground pixel count, or a negative error code. */
int64_t gar_rle_decode(const char* counts, int64_t len, int h, int w, uint8_t* mask);

/* Decode-step helpers (HF GenerationMixin greedy loop, modeling_gar.py:418-426):
 * embedding gather for the just-sampled tokens; argmax over logits with first-index tie break writing
 * out_tokens[b*out_stride + step_dev[0]] and cur_tokens[b]; device-side counters (position / step) so one captured
 * hipGraph replays for every token. */
int gar_embed_lookup(int dtype, const int64_t* tokens, const void* E, void* out, int B, int C, int64_t vocab,
                     gar_stream_t stream);
int gar_argmax(int dtype, const void* logits, int64_t ld, int B, int V, int64_t* out_tokens, int64_t out_stride,
               const int32_t* step_dev, int64_t* cur_tokens, void* workspace,
               /* the stopping criterion of HF's greedy loop (eos_token_id may be a list, modeling_gar.py:418-426 forwards the
                * caller's GenerationConfig; demo/gar_with_mask.py:112-122) evaluated where the token is produced: eos_ids
                * int64 [n_eos] (device; entries < 0 never match), finished int32 [B] (device; -1 = running, else the step
                * — column of out_tokens — at which the row FIRST produced an eos id; latched), done_count int32 [1] (device;
                * number of latched rows). All three may be NULL (n_eos = 0): plain argmax. */
               const int64_t* eos_ids, int n_eos, int32_t* finished, int32_t* done_count, gar_stream_t stream);
int64_t gar_argmax_workspace(int B, int V);
int gar_counter_add(int32_t* counters, int n, int delta, gar_stream_t stream);

/* The input checks of the reference's generate() without a host sync: image-token count vs feature rows (ValueError,
 * modeling_perception_lm.py:309-315), crop-token span length vs P*P (the splice of modeling_gar.py:404-411 would change
 * the sequence length), crop token present but no bbox (KeyError, modeling_gar.py:366), ids outside [0, vocab).
 * counts / spans: outputs of gar_placeholder_scan; has_box[b]: bit c set iff sample b has a bbox for crop token c
 * (counts NULL, or spans and has_box NULL: that group of checks is skipped — a text-only prompt).
 * ORs bits 1 / 2 / 4 / 8 into flags[0] (device int32, not cleared here).
 * attn_mask (nullable, ABI v10): uint8 / bool [B, S] generation mask (modeling_gar.py:418-426 forwards `attention_mask` to HF's
 * generate); a row that is not 0...01...1 — not LEFT-padded: HF would continue it after its padding — sets bit 16. */
int gar_input_check(const int64_t* input_ids, int B, int S, int64_t vocab, const int32_t* counts, int n_rows,
                    const int32_t* spans, int n_crop, int span_len, const int32_t* has_box, int32_t* flags,
                    const uint8_t* attn_mask, gar_stream_t stream);

/* ---------------------------------------------------------------------------------------------------------------------------
 * EXTRAS — not on the path BASELINE.json's north_star names (greedy decode; every reference caller passes do_sample=False,
 * demo/gar_with_mask.py:115-121). Kept because the reference forwards ANY GenerationConfig to HF's generate
 * (modeling_gar.py:418-426); nothing in the timed benchmark path calls into this section.
 * ------------------------------------------------------------------------------------------------------------------------- */
/* do_sample = True (ABI 13): HF's TemperatureLogitsWarper -> TopKLogitsWarper -> TopPLogitsWarper -> one multinomial draw
 * (transformers GenerationMixin._sample; the reference forwards the caller's GenerationConfig, modeling_gar.py:418-426), on the
 * device and hipGraph-replayable: params_dev float [3] = {temperature > 0, top_p in (0, 1] (1 = off), top_k (0 or >= V = off)},
 * seed_dev int64 [1]; the draw of (row b, step step_dev[0]) is Philox4x32-10(key = seed, counter = (step, row_offset + b, 0, 0)) — row_offset
 * (ABI 15) = the batch row of logits[0], for callers that produce a batch's tokens in several launches (prompt chunks) —, 24 bits ->
 * u in [0, 1), and the token is the first index in vocabulary order whose running sum of kept exp((logit - max) / T) exceeds
 * u x their total (oracle/sampling.py restates kernel and RNG; tests/test_oracle_goldens.py pins its kept set against
 * transformers' warpers). Token placement and the eos latches are gar_argmax's. */
int gar_sample(int dtype, const void* logits, int64_t ld, int B, int V, int64_t* out_tokens, int64_t out_stride,
               const int32_t* step_dev, int64_t* cur_tokens, const float* params_dev, const int64_t* seed_dev,
               const int64_t* eos_ids, int n_eos, int32_t* finished, int32_t* done_count, int row_offset, gar_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GAR_HIP_H */
