"""Operator-level Python mirror of include/gar_hip.h: tensors in, raw pointers + shapes out. Nothing here computes;
every function enqueues one HIP kernel (or a fixed pair) on the current torch stream."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import hip
from .hip import (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SCALE_RES, EPI_NONE, EPI_PATCH_POS, EPI_RES, EPI_SWIGLU,
                  GemmParams, check, dtype_code, lib, ptr, stream)


KERNEL_TIMERS = None     # bench.py sets this to a list to time launches with HIP events: (kind, flops, bytes, e0, e1)
KERNEL_PHASE = ""        # "decode:" while GARModel.generate_finish's loop enqueues (prefix of `kind`: the same skinny GEMM
                         # kernels serve the prompt phase's B-row tail and the decode steps, priced separately by bench.py)


def _timed(kind: str, nbytes: float, call, flops: float = 0.0):
    """Run one launch; when bench.py collects timers, bracket it with HIP events on the launch stream."""
    prof = KERNEL_TIMERS
    if prof is not None and not torch.cuda.is_current_stream_capturing():
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call()
        e1.record()
        prof.append((KERNEL_PHASE + kind, float(flops), float(nbytes), e0, e1))
    else:
        call()


def _chk(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise hip.GarError(f"{name}: tensor must live on the GPU")
    if not t.is_contiguous():
        raise hip.GarError(f"{name}: tensor must be contiguous")


def gemm(a: torch.Tensor, w: torch.Tensor, out: torch.Tensor, epilogue: int = EPI_NONE, bias=None, residual=None,
         gamma=None, pos=None, tokens_in: int = 0, tokens_out: int = 0, token_offset: int = 0, norm_w=None,
         norm_eps: float = 0.0, partial: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
         row_stats: Optional[torch.Tensor] = None, norm_folded: bool = False):
    """out[M, N(/2)] = epilogue(a[M,K] @ w[N,K]^T). `a`, `out`, `residual` may be row-strided 2-D views.
    ``partial`` fp32 [split_k, M, N] (decode GEMMs, EPI_NONE): the K range is cut into split_k slices whose products go
    there instead of ``out`` (pass ``out=None``); ``splitk_residual_rmsnorm`` reduces them.
    ``norm_folded`` (bf16, M <= 64): ``w`` carries an RMSNorm's gain (W diag(g)); the kernel takes rsqrt(mean(a^2) + norm_eps)
    per row from the tiles of ``a`` it streams and scales the accumulator rows — no norm launch in front of the GEMM."""
    assert a.dim() == 2
    M, K = a.shape
    p = GemmParams()
    assert w.dim() == 2 and w.shape[1] == K and w.stride(1) == 1, (a.shape, w.shape)
    N = w.shape[0]
    if partial is not None:
        assert out is None and partial.dtype == torch.float32 and partial.is_contiguous() and partial.shape[1:] == (M, N)
        p.split_k, p.partial = partial.shape[0], ptr(partial)
        out = partial
    assert a.stride(1) == 1 and out.stride(-1) == 1
    p.A, p.lda = ptr(a), a.stride(0)
    p.W, p.ldw = ptr(w), w.stride(0)
    p.C = ptr(out)
    p.ldc = out.stride(-2) if out.dim() >= 2 else out.shape[-1]
    p.M, p.N, p.K = M, N, K
    p.epilogue = epilogue
    p.bias = ptr(bias)
    p.residual = ptr(residual)
    p.ldr = residual.stride(-2) if residual is not None else 0
    p.gamma = ptr(gamma)
    p.pos = ptr(pos)
    p.tokens_in, p.tokens_out, p.token_offset = tokens_in, tokens_out, token_offset
    p.norm_w, p.norm_eps = ptr(norm_w), norm_eps
    p.norm_folded = 1 if norm_folded else 0
    # folded norm (include/gar_hip.h): row_scale [M] fp32 multiplies the accumulator rows; row_stats [M, ceil(N/64), 2] fp32
    # receives (sum, sum of squares) of the rounded outputs per 64-column strip
    if row_scale is not None:
        assert row_scale.dtype == torch.float32 and row_scale.is_contiguous() and row_scale.numel() >= M
    if row_stats is not None:
        assert row_stats.dtype == torch.float32 and row_stats.is_contiguous() and \
            tuple(row_stats.shape) == (M, (N + 63) // 64, 2)
    p.row_scale, p.row_stats = ptr(row_scale), ptr(row_stats)
    prof = KERNEL_TIMERS
    if prof is not None and not torch.cuda.is_current_stream_capturing():
        # HIP events on the launch stream around this one kernel (bench.py roofline; never inside graph capture)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(lib(a.dtype).gar_gemm(dtype_code(a.dtype), C.byref(p), stream()), "gar_gemm")
        e1.record()
        # M <= 64: the weight-streaming skinny kernels (decode GEMVs, lm_head, the pruned prefill tail) — HBM-bound, priced on
        # bytes; everything larger is the MFMA-bound tile GEMM family
        kind = ("gemm_skinny" if M <= 64 else "gemm_tile") + ("_f32" if a.dtype == torch.float32 else "_bf16")      # "_bf16": the 16-bit kernels (bf16, or fp16 in the twin library)
        prof.append((KERNEL_PHASE + kind, 2.0 * M * N * K, (M * K + N * K + M * N) * a.element_size(), e0, e1))
        return out
    check(lib(a.dtype).gar_gemm(dtype_code(a.dtype), C.byref(p), stream()), "gar_gemm")
    return out


def tile_gemm_takes(M: int, N: int, K: int = 64, lda: int = 0, ldc: int = 0, ldr: int = 0, epilogue: int = EPI_NONE,
                    row_scale: bool = False, row_stats: bool = False) -> bool:
    """True when a bf16 [M, K] x [N, K]^T problem runs on the persistent 256 x 256 tile GEMM (csrc/gemm_pp.hip) — the kernel
    whose epilogues carry the folded norms (row_scale / row_stats). The LIBRARY's predicate (gar_gemm_tile_takes): tile count,
    N >= 256, N % 8, row pitches % 8, operands < 4 GiB, epilogue / row_scale pairing. Pointer alignment is the one condition a
    shape-only question cannot carry: torch allocations are 256-byte aligned and the host only slices them at row boundaries
    of pitches that are multiples of 8 elements."""
    p = GemmParams()
    al = 256                                           # a stand-in, 16-byte aligned address for every operand the epilogue names
    p.A, p.lda, p.W, p.ldw, p.C, p.ldc = al, lda or K, al, K, al, ldc or (N // 2 if epilogue == EPI_SWIGLU else N)
    p.M, p.N, p.K, p.epilogue = M, N, K, epilogue
    if epilogue in (EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_SCALE_RES, hip.EPI_QKV_ROPE):
        p.bias = al
    if epilogue in (EPI_RES, EPI_BIAS_SCALE_RES):
        p.residual, p.ldr = al, ldr or N
    if epilogue == EPI_BIAS_SCALE_RES:
        p.gamma = al
    if epilogue == hip.EPI_QKV_ROPE:                   # the compact-table form with head-major v (what the folded path runs)
        p.qkv_q = p.qkv_k = p.qkv_v = p.qkv_sin = al
    if row_scale:
        p.row_scale = al
    if row_stats:
        p.row_stats = al
    return bool(lib().gar_gemm_tile_takes(hip.GAR_BF16, C.byref(p)))


def splitk_residual_rmsnorm(partial: torch.Tensor, h: torch.Tensor, w: Optional[torch.Tensor], eps: float,
                            out: Optional[torch.Tensor] = None):
    """h += sum over the K slices of ``partial`` [split_k, M, D] (one bf16 rounding, like EPI_RES); ``out`` = RMSNorm(h; w)
    when given (the next layer's input_layernorm / the final norm)."""
    S, M, D = partial.shape
    assert h.is_contiguous() and h.shape == (M, D) and (out is None or (out.is_contiguous() and out.shape == (M, D)))
    _timed("splitk_reduce", partial.numel() * 4 + (2 + (out is not None)) * M * D * h.element_size(), lambda: check(
        lib(h.dtype).gar_splitk_residual_rmsnorm(dtype_code(h.dtype), ptr(partial), S, ptr(h), ptr(w), ptr(out), M, D, eps,
                                          stream()), "gar_splitk_residual_rmsnorm"))
    return h


def mask_decode(mask: torch.Tensor, out: torch.Tensor, prompt_numbers: int):
    """out = (clamp(round((mask + 1) / 2 * 255), 0, P) != P) in the tensor's dtype (modeling_gar.py:315-327)."""
    _chk(mask, "mask")
    assert out.shape == mask.shape and out.dtype == mask.dtype and out.is_contiguous()
    check(lib(mask.dtype).gar_mask_decode(dtype_code(mask.dtype), ptr(mask), ptr(out), mask.numel(), prompt_numbers, stream()),
          "gar_mask_decode")
    return out


def patch_embed_k(img: int, patch: int) -> int:
    """K of the gather-ordered patch-embed weight (0: the gather form is not built for this patch size)."""
    return int(lib().gar_patch_embed_k(img, patch))


def patch_embed(pixel: torch.Tensor, maskbin: torch.Tensor, w_gather: torch.Tensor, pos: torch.Tensor, x: torch.Tensor,
                patch: int, token_offset: int) -> bool:
    """x[t, token_offset + p, :] = patch-embed(pixel) + mask-embed(maskbin) + pos, the patches DMA'd from the image tiles
    [T, 3, img, img] into LDS by the tile GEMM (no im2col matrix). False when the library does not take the shape on this
    path (nothing launched; the caller keeps patch_im2col + gemm(EPI_PATCH_POS))."""
    _chk(pixel, "pixel")
    _chk(maskbin, "maskbin")
    T, c, img, _ = pixel.shape
    assert c == 3 and maskbin.shape == pixel.shape and x.dim() == 3 and x.is_contiguous() and x.shape[0] == T
    D = x.shape[2]
    assert w_gather.is_contiguous() and w_gather.shape[0] == D and pos.is_contiguous() and pos.shape[1] == D
    prof = KERNEL_TIMERS
    timed = prof is not None and not torch.cuda.is_current_stream_capturing()
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib(pixel.dtype).gar_patch_embed(dtype_code(pixel.dtype), ptr(pixel), ptr(maskbin), ptr(w_gather), ptr(pos), ptr(x), T, img,
                               patch, D, x.shape[1], token_offset, stream())
    if rc == hip.ERR_UNSUPPORTED:
        return False
    check(rc, "gar_patch_embed")
    if timed:
        e1.record()
        g = img // patch
        M, K = T * g * g, 6 * patch * patch       # algorithmic K (the gather's zero-weight slots are not counted)
        prof.append(("gemm_tile_bf16", 2.0 * M * D * K, (T * 6 * img * img + D * K + M * D) * 2, e0, e1))
    return True


def patch_im2col(pixel: torch.Tensor, mask: Optional[torch.Tensor], out: torch.Tensor, patch: int, prompt_numbers: int):
    _chk(pixel, "pixel")
    T, c, img, _ = pixel.shape
    assert c == 3 and (mask is None or mask.shape == pixel.shape)
    check(lib(pixel.dtype).gar_patch_im2col(dtype_code(pixel.dtype), ptr(pixel), ptr(mask), ptr(out), T, img, patch, out.shape[-1],
                                 prompt_numbers, stream()), "gar_patch_im2col")
    return out


def cls_pos_fill(x: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor):
    T, tokens, D = x.shape
    check(lib(x.dtype).gar_cls_pos_fill(dtype_code(x.dtype), ptr(x), ptr(cls), ptr(pos), T, tokens, D, stream()),
          "gar_cls_pos_fill")


def tokens_add(x: torch.Tensor, add: torch.Tensor, token_offset: int):
    """x[t, token_offset + p, :] += add[t, p, :] (modeling_perception_lm.py:195-196, ``x + mask_embeds...``)."""
    T, tokens_out, D = x.shape
    assert x.is_contiguous() and add.is_contiguous() and add.dtype == x.dtype and add.shape[0] == T and add.shape[2] == D
    check(lib(x.dtype).gar_tokens_add(dtype_code(x.dtype), ptr(x), ptr(add), T, add.shape[1], tokens_out, token_offset, D, stream()),
          "gar_tokens_add")


def _rows(x):
    """(M, D, row stride) of a 2-D row-strided view or a contiguous [..., D] tensor."""
    D = x.shape[-1]
    assert x.stride(-1) == 1
    if x.dim() == 2:
        return x.shape[0], D, x.stride(0)
    assert x.is_contiguous()
    return x.numel() // D, D, D


def layernorm(x, w, b, eps: float, out=None):
    out = x if out is None else out
    M, D, ldx = _rows(x)
    _, _, ldy = _rows(out)
    check(lib(x.dtype).gar_layernorm(dtype_code(x.dtype), ptr(x), ptr(out), ptr(w), ptr(b), M, D, ldx, ldy, eps, stream()),
          "gar_layernorm")
    return out


def rmsnorm(x, w, eps: float, out=None):
    out = x if out is None else out
    M, D, ldx = _rows(x)
    _, _, ldy = _rows(out)
    check(lib(x.dtype).gar_rmsnorm(dtype_code(x.dtype), ptr(x), ptr(out), ptr(w), M, D, ldx, ldy, eps, stream()),
          "gar_rmsnorm")
    return out


def row_rstd(x, eps: float, rms: bool, out):
    """out[m] = rsqrt(var(x[m]) + eps) (LayerNorm) or rsqrt(mean(x[m]^2) + eps) (RMSNorm), fp32 [M]: the statistics half of a
    norm whose scaling half is folded into the next GEMM (``gemm(..., row_scale=out)``)."""
    M, D, ldx = _rows(x)
    assert out.dtype == torch.float32 and out.is_contiguous() and out.numel() >= M
    check(lib(x.dtype).gar_row_rstd(dtype_code(x.dtype), ptr(x), M, D, ldx, eps, int(rms), ptr(out), stream()), "gar_row_rstd")
    return out


def row_stats_finalize(stats, D: int, eps: float, rms: bool, out):
    """stats [M, strips, 2] fp32 (sum, sum of squares per 64-column strip, written by a producer GEMM's ``row_stats=``)
    -> out[m] = rstd of row m."""
    M, strips, two = stats.shape
    assert two == 2 and stats.dtype == torch.float32 and stats.is_contiguous() and out.dtype == torch.float32
    check(lib().gar_row_stats_finalize(ptr(stats), M, strips, D, eps, int(rms), ptr(out), stream()),
          "gar_row_stats_finalize")
    return out


def vit_qkv_post(qkv, sin, cos, Q, K, Vt, T, N, npt, H, hd, Npad, q_scale):
    check(lib(qkv.dtype).gar_vit_qkv_post(dtype_code(qkv.dtype), ptr(qkv), ptr(sin), ptr(cos), ptr(Q), ptr(K), ptr(Vt), T, N, npt,
                                 H, hd, Npad, q_scale, stream()), "gar_vit_qkv_post")


_SINCOS_CACHE = {}


def _compact_sincos(sin: torch.Tensor, cos: torch.Tensor):
    """timm's RotaryEmbeddingCat repeats every angle for both elements of a rotated pair (repeat_interleave(2),
    SURVEY.md A.1); when the tables have that form return the compact (sin, cos)-pair table the library's fast QKV_ROPE
    epilogue reads (include/gar_hip.h: qkv_cos == NULL), else None. Cached per table pair."""
    key = (sin.data_ptr(), cos.data_ptr(), tuple(sin.shape))
    hit = _SINCOS_CACHE.get(key)
    if hit is None:
        ok = (sin.dtype == torch.float32 and cos.dtype == torch.float32 and sin.shape[-1] % 8 == 0 and
              bool(torch.equal(sin[:, 0::2], sin[:, 1::2])) and bool(torch.equal(cos[:, 0::2], cos[:, 1::2])))
        if ok:      # row 0 = the identity (sin, cos) = (0, 1): what the un-rotated prefix rows and the v columns read (gar_hip.h, ABI 14)
            pairs = torch.stack([sin[:, 0::2], cos[:, 0::2]], dim=-1)
            ident = torch.stack([torch.zeros_like(pairs[:1, :, 0]), torch.ones_like(pairs[:1, :, 1])], dim=-1)
            pairs = torch.cat([ident, pairs], 0).contiguous()
        hit = (pairs if ok else False, sin, cos)   # keeps the key's tensors alive
        _SINCOS_CACHE[key] = hit
    return hit[0] if hit[0] is not False else None


def gemm_qkv_rope(a, w, bias, v_out, Q, K, sin, cos, heads, hd, tokens, tokens_pad, prefix, q_scale,
                  compact: bool = True, V: Optional[torch.Tensor] = None,
                  row_scale: Optional[torch.Tensor] = None) -> bool:
    """qkv GEMM with the front half of timm AttentionRope fused (GAR_EPI_QKV_ROPE): q / k are rotated, scaled and written
    straight into Q / K [tiles, heads, tokens_pad, hd]; v goes row-major to ``v_out`` [M, heads*hd]. Returns False when the
    library does not take this shape / dtype on the fused path (caller keeps gemm + vit_qkv_post)."""
    M, Kd = a.shape
    N = w.shape[0]
    p = GemmParams()
    p.A, p.lda = ptr(a), a.stride(0)
    p.W, p.ldw = ptr(w), w.stride(0)
    p.C, p.ldc = ptr(v_out), v_out.stride(0)
    p.M, p.N, p.K = M, N, Kd
    p.epilogue = hip.EPI_QKV_ROPE
    p.bias = ptr(bias)
    # compact=False forces the general form (independent sin / cos tables, workgroup-level epilogue) that tables without the
    # pair structure take anyway
    # (the compact table goes with the per-wave epilogue, which writes v head-major: without V= the general form runs)
    sc = _compact_sincos(sin, cos) if (compact and V is not None) else None
    if sc is not None:         # (sin, cos) pairs [tokens, hd/2, 2]: half the table bytes, the barrier-free per-wave epilogue
        p.qkv_q, p.qkv_k, p.qkv_sin, p.qkv_cos = ptr(Q), ptr(K), ptr(sc), None
    else:
        p.qkv_q, p.qkv_k, p.qkv_sin, p.qkv_cos = ptr(Q), ptr(K), ptr(sin), ptr(cos)
    p.qkv_heads, p.qkv_head_dim, p.qkv_tokens, p.qkv_tokens_pad, p.qkv_prefix = heads, hd, tokens, tokens_pad, prefix
    p.qkv_q_scale = q_scale
    # V [tiles, heads, tokens_pad, hd]: v leaves the GEMM head-major like k (attention(..., v_row_major=True) reads it in
    # place); v_out is then not written
    p.qkv_v = ptr(V)
    p.row_scale = ptr(row_scale)           # folded LayerNorm: a = the residual stream itself, w / bias the folded pair
    prof = KERNEL_TIMERS
    timed = prof is not None and not torch.cuda.is_current_stream_capturing()
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib(a.dtype).gar_gemm(dtype_code(a.dtype), C.byref(p), stream())
    if rc == hip.ERR_UNSUPPORTED:
        return False
    check(rc, "gar_gemm(QKV_ROPE)")
    if timed:
        e1.record()
        prof.append(("gemm_tile_bf16", 2.0 * M * N * Kd, (M * Kd + N * Kd + M * N) * a.element_size(), e0, e1))
    return True


def llm_qkv_weight_order(hd: int, Hq: int, Hkv: int) -> torch.Tensor:
    """Row order of the fused Llama qkv weight that GAR_EPI_QKV_ROPE_LLM expects (include/gar_hip.h): inside every q / k head,
    64-row strip j holds dims [32j, 32j+32) then [hd/2 + 32j, hd/2 + 32j + 32) — a rotation's two halves share a strip;
    v heads keep their order. Identity for head_dim 64. ``w[order]`` is the weight to pass."""
    half = hd // 2
    per_head = torch.cat([torch.cat([torch.arange(32 * j, 32 * j + 32), torch.arange(half + 32 * j, half + 32 * j + 32)])
                          for j in range(hd // 64)])
    qk = (torch.arange(Hq + Hkv)[:, None] * hd + per_head[None, :]).reshape(-1)
    return torch.cat([qk, torch.arange((Hq + Hkv) * hd, (Hq + 2 * Hkv) * hd)])


def gemm_qkv_rope_llm(a, w, Q, Kc, Vc, cos, sin, B, S, Spad, Hq, Hkv, hd, Smax, pos0, pos_dev, q_scale, left_pad=None,
                      row_scale: Optional[torch.Tensor] = None) -> bool:
    """Llama qkv GEMM with gar_llm_qkv_post fused into its epilogue (GAR_EPI_QKV_ROPE_LLM): ``a`` [B*S, C], ``w`` in
    :func:`llm_qkv_weight_order`; q goes rotated and scaled to ``Q`` [B, Hq, Spad, hd], k (rotated) and v to rows
    pos0 .. pos0 + S - 1 of ``Kc`` / ``Vc`` [B, Hkv, Smax, hd]. Returns False when the library does not take this shape /
    dtype on the fused path (caller keeps gemm + llm_qkv_post with the natural weight order)."""
    M, Kd = a.shape
    N = w.shape[0]
    p = GemmParams()
    p.A, p.lda = ptr(a), a.stride(0)
    p.W, p.ldw = ptr(w), w.stride(0)
    p.C, p.ldc = None, 0
    p.M, p.N, p.K = M, N, Kd
    p.epilogue = hip.EPI_QKV_ROPE_LLM
    p.qkv_q, p.qkv_k, p.qkv_v, p.qkv_sin, p.qkv_cos = ptr(Q), ptr(Kc), ptr(Vc), ptr(sin), ptr(cos)
    p.qkv_heads, p.qkv_head_dim, p.qkv_tokens, p.qkv_tokens_pad = Hq, hd, S, Spad
    p.qkv_kv_heads, p.qkv_kv_stride, p.qkv_pos0 = Hkv, Smax, pos0
    p.qkv_pos_dev, p.qkv_left_pad = ptr(pos_dev), ptr(left_pad)
    p.qkv_q_scale = q_scale
    p.row_scale = ptr(row_scale)
    prof = KERNEL_TIMERS
    timed = prof is not None and not torch.cuda.is_current_stream_capturing()
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib(a.dtype).gar_gemm(dtype_code(a.dtype), C.byref(p), stream())
    if rc == hip.ERR_UNSUPPORTED:
        return False
    check(rc, "gar_gemm(QKV_ROPE_LLM)")
    if timed:
        e1.record()
        prof.append(("gemm_tile_bf16", 2.0 * M * N * Kd, (M * Kd + N * Kd + M * N) * a.element_size(), e0, e1))
    return True


def vit_v_transpose(v, Vt, T, N, H, hd, Npad):
    check(lib(v.dtype).gar_vit_v_transpose(dtype_code(v.dtype), ptr(v), ptr(Vt), T, N, H, hd, Npad, stream()),
          "gar_vit_v_transpose")


def llm_qkv_post(qkv, cos, sin, Q, Kc, Vc, B, S, Spad, Hq, Hkv, hd, Smax, pos0, pos_dev, q_scale, left_pad=None,
                 strip_order: bool = False):
    """``Kc``, ``Vc`` [B, Hkv, Smax, hd]. ``left_pad`` int32 [B] (device) or None: first real row of each sequence of a
    left-padded batch. ``strip_order``: the q / k head columns of ``qkv`` are in :func:`llm_qkv_weight_order`'s order."""
    check(lib(qkv.dtype).gar_llm_qkv_post(dtype_code(qkv.dtype), ptr(qkv), ptr(cos), ptr(sin), ptr(Q), ptr(Kc), ptr(Vc), B, S,
                                 Spad, Hq, Hkv, hd, Smax, pos0, ptr(pos_dev), ptr(left_pad), q_scale, int(strip_order),
                                 stream()), "gar_llm_qkv_post")


def attention(Q, K, Vt, O, B, Hq, Hkv, hd, q_len, q_pad, kv_len, kv_stride, causal, kv_len_dev=None,
              v_row_major: bool = False, kv_start=None, kv_prefix: int = 0):
    """``v_row_major``: ``Vt`` is V [B, Hkv, kv_stride, hd] (K's layout: the fused qkv GEMMs' v output, the Llama KV cache)
    instead of its transpose.
    ``kv_start`` int32 [B] (device) or None: first visible kv row per sequence (left-padded batch).
    ``kv_prefix`` = 1 (v_row_major, non-causal): kv row 0 is folded into the softmax's initial state (ViT cls token)."""
    # algorithmic flops: QK^T + PV over the (query, key) pairs that exist — q_len x kv_len, or the causal triangle
    pairs = (q_len * (q_len + 1) // 2 + q_len * (kv_len - q_len)) if causal else q_len * kv_len
    kind = "attn_causal" if causal else "attn_full"
    if v_row_major:
        _timed(kind, 0.0, lambda: check(
            lib(Q.dtype).gar_attention_vrow(dtype_code(Q.dtype), ptr(Q), ptr(K), ptr(Vt), ptr(O), B, Hq, Hkv, hd, q_len, q_pad,
                                     kv_len, kv_stride, int(causal), ptr(kv_len_dev), ptr(kv_start), int(kv_prefix),
                                     stream()), "gar_attention_vrow"), flops=4.0 * B * Hq * hd * pairs)
        return
    _timed(kind, 0.0, lambda: check(
        lib(Q.dtype).gar_attention(dtype_code(Q.dtype), ptr(Q), ptr(K), ptr(Vt), ptr(O), B, Hq, Hkv, hd, q_len, q_pad,
                            kv_len, kv_stride, int(causal), ptr(kv_len_dev), ptr(kv_start), stream()), "gar_attention"),
        flops=4.0 * B * Hq * hd * pairs)


def attention_decode(q, Kc, Vc, O, B, Hq, Hkv, hd, Smax, kv_len_dev, max_splits, workspace, kv_start=None,
                     q_stride: int = 0):
    """``q_stride`` (elements between the query rows of consecutive (b, head); 0 = hd, the packed [B, Hq, hd] form): with
    ``q = Q[:, :, S - 1]`` of a prefill's Q [B, Hq, Spad, hd] and ``q_stride = Spad * hd`` the last prompt row is read in place."""
    # (bytes: the kv length lives in device memory; bench.py prices a launch at B * Hkv * kv_len * hd * 2 tensors * 2 B)
    _timed("attn_decode", 0.0, lambda: check(
        lib(q.dtype).gar_attention_decode(dtype_code(q.dtype), ptr(q), int(q_stride), ptr(Kc), ptr(Vc), ptr(O), B, Hq, Hkv, hd, Smax,
                                   ptr(kv_len_dev), ptr(kv_start), max_splits, ptr(workspace), stream()),
        "gar_attention_decode"))


def attention_decode_qkv(qkv, cos, sin, Kc, Vc, O, B, Hq, Hkv, hd, Smax, pos_dev, q_scale, max_splits, workspace,
                         left_pad=None, strip_order: bool = False) -> bool:
    """llm_qkv_post (S = 1) + attention_decode in one launch: ``qkv`` [B, (Hq + 2 Hkv) hd] raw GEMM output of the step, the
    new key / value rows are appended to ``Kc`` / ``Vc`` at row pos_dev[0]. False (nothing launched) in parity mode."""
    prof = KERNEL_TIMERS
    timed = prof is not None and not torch.cuda.is_current_stream_capturing()
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    rc = lib(qkv.dtype).gar_attention_decode_qkv(dtype_code(qkv.dtype), ptr(qkv), ptr(cos), ptr(sin), ptr(Kc), ptr(Vc), ptr(O), B, Hq,
                                        Hkv, hd, Smax, ptr(pos_dev), ptr(left_pad), q_scale, int(strip_order), max_splits,
                                        ptr(workspace), stream())
    if rc == hip.ERR_UNSUPPORTED:
        return False
    check(rc, "gar_attention_decode_qkv")
    if timed:
        e1.record()
        prof.append((KERNEL_PHASE + "attn_decode", 0.0, 0.0, e0, e1))
    return True


def attention_decode_workspace(B, Hq, hd, max_splits) -> int:
    return int(lib().gar_attention_decode_workspace(B, Hq, hd, max_splits))


def pool2x2(x, y, g: int, in_tile_tokens: int = 0, in_token_offset: int = 0):
    """x [T, in_tile_tokens, C] (tokens in_token_offset .. +g*g of each tile are the g x g grid) -> y [T, (g/2)^2, C]"""
    T, Cc = y.shape[0], y.shape[-1]
    _timed("pool2x2", (T * g * g * Cc + y.numel()) * x.element_size(), lambda: check(
        lib(x.dtype).gar_pool2x2(dtype_code(x.dtype), ptr(x), ptr(y), T, g, Cc, in_tile_tokens, in_token_offset, stream()),
        "gar_pool2x2"))
    return y


def placeholder_scan(input_ids, image_token_id, crop_ids_dev, slot, counts, spans, rank_pos=None):
    """rank_pos (optional) int32 [B, n_rows]: position of the image token of rank r (inverse of ``slot``)."""
    B, S = input_ids.shape
    assert input_ids.dtype == torch.int64 and input_ids.is_contiguous()
    check(lib().gar_placeholder_scan(ptr(input_ids), B, S, int(image_token_id), ptr(crop_ids_dev),
                                     crop_ids_dev.numel(), ptr(slot), ptr(counts), ptr(spans), ptr(rank_pos),
                                     0 if rank_pos is None else rank_pos.shape[1], stream()), "gar_placeholder_scan")


def pool_assemble(input_ids, slot, E, proj, out, tiles_per_sample, g, in_tile_tokens=0, in_token_offset=0):
    """out [B,S,C] = embedding rows, image-token rows = 2x2 mean of the projector output ``proj`` [B*tiles*in_tile_tokens, C]
    (pool2x2 + embed_assemble in one pass; the pooled features stay inside ``out``)."""
    B, S = input_ids.shape
    C_ = E.shape[1]
    n_img = B * tiles_per_sample * (g // 2) ** 2
    # algorithmic bytes: the grid rows of the projector output read once + the sequence written once
    _timed("pool_assemble", (4 * n_img * C_ + out.numel()) * out.element_size(), lambda: check(
        lib(E.dtype).gar_pool_assemble(dtype_code(E.dtype), ptr(input_ids), ptr(slot), ptr(E), ptr(proj), ptr(out), B, S, C_,
                                tiles_per_sample, g, in_tile_tokens, in_token_offset, E.shape[0], stream()),
        "gar_pool_assemble"))


def roi_replay_inplace(embeds, spans, rank_pos, jobs, n_crop, P, Cc, S, sampling_ratio=2, aligned=True):
    """batched RoI replay reading the pooled features from the image-token rows of ``embeds`` (see pool_assemble)."""
    n = jobs.numel() // 40
    _timed("roi_replay", n * (P * P + 16) * Cc * embeds.element_size(), lambda: check(
        lib(embeds.dtype).gar_roi_replay_inplace(dtype_code(embeds.dtype), ptr(embeds), ptr(spans), ptr(rank_pos), rank_pos.shape[1],
                                     ptr(jobs), n, n_crop, P, Cc, S, sampling_ratio, int(aligned), stream()),
        "gar_roi_replay_inplace"))


def embed_assemble(input_ids, slot, E, feats, out, n_feat_rows):
    B, S = input_ids.shape
    _timed("embed_assemble", 2 * out.numel() * out.element_size(), lambda: check(
        lib(E.dtype).gar_embed_assemble(dtype_code(E.dtype), ptr(input_ids), ptr(slot), ptr(E), ptr(feats), ptr(out), B, S,
                                 E.shape[1], int(n_feat_rows), E.shape[0], stream()), "gar_embed_assemble"))


def roi_replay(feats, embeds, spans, crop_index, first_tile, ncw, nch, P, Cc, S, roi, spatial_scale,
               sampling_ratio=2, aligned=True):
    check(lib(feats.dtype).gar_roi_replay(dtype_code(feats.dtype), ptr(feats), ptr(embeds), ptr(spans), crop_index, first_tile, ncw,
                               nch, P, Cc, S, roi[0], roi[1], roi[2], roi[3], spatial_scale, sampling_ratio,
                               int(aligned), stream()), "gar_roi_replay")


ROI_JOB_DTYPE = [("sample", "<i4"), ("crop_index", "<i4"), ("first_tile", "<i4"), ("ncw", "<i4"), ("nch", "<i4"),
                 ("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("spatial_scale", "<f4")]   # gar_roi_job


def roi_replay_batched(feats, embeds, spans, jobs, n_crop, tiles_per_sample, P, Cc, S, sampling_ratio=2, aligned=True):
    """jobs: uint8 device tensor holding n packed ``gar_roi_job`` records (see ``roi_jobs_tensor``)."""
    n = jobs.numel() // 40
    # algorithmic bytes per crop token: P*P*C written + <= 16 map cells * C read
    _timed("roi_replay", n * (P * P + 16) * Cc * feats.element_size(), lambda: check(
        lib(feats.dtype).gar_roi_replay_batched(dtype_code(feats.dtype), ptr(feats), ptr(embeds), ptr(spans), ptr(jobs), n,
                                     n_crop, tiles_per_sample, P, Cc, S, sampling_ratio, int(aligned), stream()),
        "gar_roi_replay_batched"))


def roi_jobs_tensor(jobs, device):
    """[(sample, crop_index, first_tile, ncw, nch, x1, y1, x2, y2, spatial_scale)] -> packed device records."""
    import numpy as np
    arr = np.array([tuple(j) for j in jobs], dtype=ROI_JOB_DTYPE)
    assert arr.dtype.itemsize == 40
    return torch.from_numpy(arr.view(np.uint8).copy()).to(device)


def resize_bicubic_tiles(src_u8, tmp, out, ts, ncw, tile0, xt, yt, mean, std):
    """src_u8 [H,W,3] uint8 (device) -> normalised tiles written into out [tiles,3,ts,ts]; xt / yt = (first, count,
    weights[n_out,kmax]) device tap tables of the two passes; tmp: fp32 scratch of >= 3*H*Wout elements."""
    H, W, _ = src_u8.shape
    Wout, Hout = xt[2].shape[0], yt[2].shape[0]
    check(lib().gar_resize_bicubic_h(ptr(src_u8), ptr(tmp), H, W, Wout, ptr(xt[0]), ptr(xt[1]), ptr(xt[2]),
                                     xt[2].shape[1], stream()), "gar_resize_bicubic_h")
    check(lib(out.dtype).gar_resize_bicubic_v_tiles(dtype_code(out.dtype), ptr(tmp), ptr(out), H, Wout, Hout, ts, ncw, tile0,
                                           ptr(yt[0]), ptr(yt[1]), ptr(yt[2]), yt[2].shape[1], mean, std, stream()),
          "gar_resize_bicubic_v_tiles")


def resize_nearest_tiles(src_u8, out, ts, ncw, tile0, xi, yi, mean, std):
    H, W, _ = src_u8.shape
    check(lib(out.dtype).gar_resize_nearest_tiles(dtype_code(out.dtype), ptr(src_u8), ptr(out), H, W, yi.numel(), xi.numel(), ts,
                                         ncw, tile0, ptr(xi), ptr(yi), mean, std, stream()), "gar_resize_nearest_tiles")


def embed_lookup(tokens, E, out):
    check(lib(E.dtype).gar_embed_lookup(dtype_code(E.dtype), ptr(tokens), ptr(E), ptr(out), tokens.numel(), E.shape[1],
                                 E.shape[0], stream()), "gar_embed_lookup")


def argmax(logits, V, out_tokens, out_stride, step_dev, cur_tokens, workspace, eos_ids=None, finished=None, done_count=None):
    """``eos_ids`` int64 [n] (device, entries < 0 never match), ``finished`` int32 [B] (-1 = running; the kernel latches the step
    at which a row first produced an eos id), ``done_count`` int32 [1] (number of latched rows): the greedy loop's stopping
    criterion evaluated on the device (all optional)."""
    B = logits.shape[0]
    assert eos_ids is None or (eos_ids.dtype == torch.int64 and eos_ids.is_contiguous())
    assert finished is None or (eos_ids is not None and finished.dtype == torch.int32 and finished.is_contiguous()
                                and finished.numel() == B)
    check(lib(logits.dtype).gar_argmax(dtype_code(logits.dtype), ptr(logits), logits.stride(0), B, V, ptr(out_tokens), out_stride,
                           ptr(step_dev), ptr(cur_tokens), ptr(workspace), ptr(eos_ids),
                           0 if eos_ids is None else eos_ids.numel(), ptr(finished), ptr(done_count), stream()), "gar_argmax")


def sample(logits, V, out_tokens, out_stride, step_dev, cur_tokens, params, seed, eos_ids=None, finished=None, done_count=None,
           row_offset: int = 0):
    """do_sample = True: temperature / top-k / top-p + one Philox draw per row (``gar_sample``). ``params`` float32 [3] (device):
    temperature, top_p, top_k; ``seed`` int64 [1] (device). Token placement and eos latches as :func:`argmax`. ``row_offset``: the
    batch row of ``logits[0]`` — the draw of row b is counted by (step, row_offset + b), so a batch whose first tokens come out of several
    prompt chunks draws what the un-chunked batch draws (ABI 15)."""
    B = logits.shape[0]
    assert params.dtype == torch.float32 and params.numel() >= 3 and params.is_contiguous()
    assert seed.dtype == torch.int64 and seed.numel() >= 1
    assert eos_ids is None or (eos_ids.dtype == torch.int64 and eos_ids.is_contiguous())
    assert finished is None or (eos_ids is not None and finished.dtype == torch.int32 and finished.is_contiguous()
                                and finished.numel() == B)
    check(lib(logits.dtype).gar_sample(dtype_code(logits.dtype), ptr(logits), logits.stride(0), B, V, ptr(out_tokens), out_stride,
                                       ptr(step_dev), ptr(cur_tokens), ptr(params), ptr(seed), ptr(eos_ids),
                                       0 if eos_ids is None else eos_ids.numel(), ptr(finished), ptr(done_count), int(row_offset),
                                       stream()),
          "gar_sample")


def argmax_workspace(B, V) -> int:
    return int(lib().gar_argmax_workspace(B, V))


def input_check(input_ids, vocab: int, counts, n_rows: int, spans, span_len: int, has_box, flags, attn_mask=None):
    """device-side input checks of generate(validate=False): ORs INPUT_* bits into ``flags`` (int32 [1]). ``attn_mask``: bool /
    uint8 [B, S] generation mask; a row that is not left-padded sets INPUT_MASK_NOT_LEFT_PADDED."""
    B, S = input_ids.shape
    if attn_mask is not None:
        assert attn_mask.dtype in (torch.bool, torch.uint8) and attn_mask.is_contiguous() and tuple(attn_mask.shape) == (B, S)
    check(lib().gar_input_check(ptr(input_ids), B, S, int(vocab), ptr(counts), int(n_rows), ptr(spans),
                                0 if spans is None else spans.shape[1], int(span_len), ptr(has_box), ptr(flags), ptr(attn_mask),
                                stream()), "gar_input_check")


def counter_add(counters, delta: int):
    check(lib().gar_counter_add(ptr(counters), counters.numel(), delta, stream()), "gar_counter_add")
