"""CPU ORACLE — TEST INFRASTRUCTURE ONLY.

An fp32 CPU restatement (torch-CPU / numpy) of the reference's region-captioning hot path,
``GARModel.generate`` and everything it calls. Only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import this file; the product (``gar_amd``) never does.

Every function cites the reference lines (paths under /root/reference) it restates. Pieces whose
arithmetic lives in third-party packages that are absent from /root/reference are restated from
their published algorithms and named here with the version the reference pins:

  * timm==1.0.19 ``Eva`` / ``AttentionRope`` / ``RotaryEmbeddingCat``  -> ``pe_vit_forward``
        PARITY UNPINNED (timm is not installable here; restated from timm/models/eva.py,
        timm/layers/attention.py, timm/layers/pos_embed_sincos.py; RoPE grid indexing is
        config-switchable for that reason)
  * torchvision ``ops.roi_align`` (CPU kernel roi_align_kernel.cpp)        -> ``roi_align``
        PARITY UNPINNED (torchvision absent); pinned only by known-answer tests
        (constant map, linear ramp) and an independent C restatement (roi_align_ref.c)
  * transformers==4.56.2 ``LlamaModel`` + greedy ``GenerationMixin``        -> ``llama_*`` / ``greedy_generate``
        PINNED: checked against transformers 5.15.0 ``LlamaForCausalLM`` outputs captured in
        tests/golden (tools/make_goldens.py)
  * stock ``PerceptionLMMultiModalProjector`` / ``AdaptiveAvgPooling``        -> ``projector_forward``
        PINNED against transformers' classes (byte-identical logic to modeling_perception_lm.py:42-92)

All math is float32 on CPU; ``attn_impl="sdpa"`` only swaps the softmax(QK^T)V evaluation for torch's
fused CPU kernel (same fp32 arithmetic up to summation order) so the timed baseline is not
handicapped by a materialised S x S score matrix.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

VT = "mllm.model.vision_tower.timm_model."
PJ = "mllm.model.multi_modal_projector."
LM = "mllm.model.language_model."


# =============================================================================================
# A1/A2  mask decode + mask patch embedding          modeling_gar.py:315-328
# =============================================================================================
def decode_mask_values(global_mask_values: torch.Tensor, prompt_numbers: int) -> torch.Tensor:
    """round((m+1)/2*255) -> long -> clamp[0,P] -> (v != P) as float   (modeling_gar.py:315-327).
    The arithmetic is evaluated in the dtype of ``global_mask_values`` exactly as the reference
    does (each op rounds to that dtype)."""
    mv = torch.round((global_mask_values + 1.0) / 2.0 * 255.0).long()
    mv = torch.clamp(mv, min=0, max=prompt_numbers)
    assert mv.max() < prompt_numbers + 1 and mv.min() >= 0, f"max: {mv.max()}, min: {mv.min()}"
    return (mv != prompt_numbers).to(torch.float32)


def mask_patch_embed(binary: torch.Tensor, weight: torch.Tensor) -> torch.Tensor:
    """nn.Conv2d(3, C_v, k=s=patch, bias=False)   (modeling_gar.py:54-60,326-328)."""
    k = weight.shape[-1]
    return F.conv2d(binary, weight, bias=None, stride=k)


# =============================================================================================
# A4  PE ViT (timm Eva, "vit_pe_lang_*")              modeling_perception_lm.py:181-221 + timm [EXT]
# =============================================================================================
def rope2d_tables(vcfg) -> (torch.Tensor, torch.Tensor):
    """timm RotaryEmbeddingCat(head_dim, in_pixels=False, feat_shape=ref_feat_shape, grid_offset=1,
    grid_indexing='xy', temperature=1e4).get_embed() split into (sin, cos), each [n_patches, head_dim].
    bands = 1/T^(arange(hd/4)/(hd/4)); coords = arange(g)+offset; per axis pos = coord*bands;
    emb = cat(axis0, axis1) then repeat_interleave(2)  (SURVEY.md A.1)."""
    hd = vcfg.head_dim
    g = vcfg.grid
    nb = hd // 4
    bands = 1.0 / (vcfg.rope_temperature ** (torch.arange(0, nb, dtype=torch.int64).to(torch.float32) / nb))
    t = torch.arange(g, dtype=torch.int64).to(torch.float32) + vcfg.rope_grid_offset
    # torch.meshgrid([t_h, t_w], indexing=...) then stack(-1): with 'xy' the first component at flattened
    # position p=(row i, col j) is t[j] (x), the second t[i] (y); with 'ij' it is (t[i], t[j]).
    g0, g1 = torch.meshgrid(t, t, indexing=vcfg.rope_grid_indexing)
    grid = torch.stack([g0, g1], dim=-1)                       # [g, g, 2]
    pos = grid.unsqueeze(-1) * bands                           # [g, g, 2, nb]
    sin = pos.sin().reshape(g * g, -1).repeat_interleave(2, -1)   # [n, hd]
    cos = pos.cos().reshape(g * g, -1).repeat_interleave(2, -1)
    return sin, cos


def _rot_interleaved(x: torch.Tensor) -> torch.Tensor:
    # timm.layers.pos_embed_sincos.rot: stack([-x[..., 1::2], x[..., ::2]], -1).reshape(x.shape)
    return torch.stack([-x[..., 1::2], x[..., ::2]], -1).reshape(x.shape)


def pe_vit_forward(pixel_values: torch.Tensor, mask_embeds: Optional[torch.Tensor], W: Dict[str, torch.Tensor],
                   cfg, attn_impl: str = "eager", return_layers: bool = False):
    """custom_forward_features (modeling_perception_lm.py:181-221):
    patch_embed -> (+mask_embeds, :195-196) -> _pos_embed (cls cat, +pos, rope) -> norm_pre -> blocks -> norm(Identity).
    pixel_values [T,3,H,W]; returns [T, npt+n_patches, D]."""
    v = cfg.mllm_config.vision_config
    D, H, hd = v.embed_dim, v.num_heads, v.head_dim
    npt = 1 if (VT + "cls_token") in W else 0
    x = F.conv2d(pixel_values, W[VT + "patch_embed.proj.weight"], bias=None, stride=v.patch_size)
    x = x.flatten(2).transpose(1, 2)                                        # [T, n, D] row-major patches
    if mask_embeds is not None:
        x = x + mask_embeds.flatten(2).transpose(1, 2)                      # :195-196
    if npt:
        x = torch.cat((W[VT + "cls_token"].expand(x.shape[0], -1, -1), x), dim=1)
    x = x + W[VT + "pos_embed"]
    sin, cos = rope2d_tables(v)
    x = F.layer_norm(x, (D,), W[VT + "norm_pre.weight"], W[VT + "norm_pre.bias"], v.ln_eps)
    layers = []
    T, N, _ = x.shape
    for i in range(v.depth):
        b = f"{VT}blocks.{i}."
        h = F.layer_norm(x, (D,), W[b + "norm1.weight"], W[b + "norm1.bias"], v.ln_eps)
        qkv = F.linear(h, W[b + "attn.qkv.weight"], W[b + "attn.qkv.bias"])
        qkv = qkv.reshape(T, N, 3, H, hd).permute(2, 0, 3, 1, 4)
        q, k, vv = qkv.unbind(0)                                            # [T, H, N, hd]
        q = torch.cat([q[:, :, :npt], q[:, :, npt:] * cos + _rot_interleaved(q[:, :, npt:]) * sin], dim=2)
        k = torch.cat([k[:, :, :npt], k[:, :, npt:] * cos + _rot_interleaved(k[:, :, npt:]) * sin], dim=2)
        if attn_impl == "sdpa":
            a = F.scaled_dot_product_attention(q, k, vv)
        else:
            s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
            a = torch.softmax(s, dim=-1) @ vv
        a = a.transpose(1, 2).reshape(T, N, D)
        a = F.linear(a, W[b + "attn.proj.weight"], W[b + "attn.proj.bias"])
        x = x + W[b + "gamma_1"] * a
        h = F.layer_norm(x, (D,), W[b + "norm2.weight"], W[b + "norm2.bias"], v.ln_eps)
        h = F.linear(h, W[b + "mlp.fc1.weight"], W[b + "mlp.fc1.bias"])
        h = F.gelu(h)                                                        # nn.GELU() exact erf
        h = F.linear(h, W[b + "mlp.fc2.weight"], W[b + "mlp.fc2.bias"])
        x = x + W[b + "gamma_2"] * h
        if return_layers:
            layers.append(x.clone())
    return (x, layers) if return_layers else x


# =============================================================================================
# A5/A6  projector + 2x2 adaptive average pool       modeling_perception_lm.py:42-92
# =============================================================================================
def projector_forward(x: torch.Tensor, W: Dict[str, torch.Tensor], pooling_ratio: int = 2) -> torch.Tensor:
    """linear_1 -> GELU(erf) -> linear_2 -> NLC->NCHW -> adaptive_avg_pool2d -> NLC  (:85-92, :47-60)."""
    y = F.linear(x, W[PJ + "linear_1.weight"], W[PJ + "linear_1.bias"])
    y = F.gelu(y)
    y = F.linear(y, W[PJ + "linear_2.weight"], W[PJ + "linear_2.bias"])
    if pooling_ratio > 1:
        b, n, c = y.shape
        h = int(math.sqrt(n))
        if h * h != n:
            raise ValueError(f"num_tokens {n} is expected to be a square number")
        y = y.permute(0, 2, 1).reshape(b, -1, h, h)
        y = F.adaptive_avg_pool2d(y, (h // pooling_ratio, h // pooling_ratio))
        y = y.flatten(2).transpose(1, 2)
    return y


def get_image_features(pixel_values, mask_embeds, W, cfg, attn_impl="eager") -> torch.Tensor:
    """PerceptionLMModel.get_image_features (modeling_perception_lm.py:239-269)."""
    if pixel_values.dim() == 5:
        pixel_values = pixel_values.flatten(0, 1)
    assert pixel_values.dim() == 4
    x = pe_vit_forward(pixel_values, mask_embeds, W, cfg, attn_impl)
    if cfg.mllm_config.vision_use_cls_token:
        x = x[:, 1:, :]
    return projector_forward(x, W, cfg.mllm_config.projector_pooling_ratio)


# =============================================================================================
# A7  embedding + placeholder scatter                 modeling_gar.py:332-346, modeling_perception_lm.py:271-331
# =============================================================================================
def embed_and_scatter(input_ids: torch.Tensor, E: torch.Tensor, image_features: torch.Tensor,
                      image_token_id: int) -> torch.Tensor:
    inputs_embeds = F.embedding(input_ids, E)                                # [B,S,C]
    special = (input_ids == image_token_id)
    n_tok = int(special.sum())
    mask = special.unsqueeze(-1).expand_as(inputs_embeds)
    if inputs_embeds[mask].numel() != image_features.numel():
        raise ValueError(
            f"Image features and image tokens do not match: tokens: {n_tok}, "
            f"features {image_features.shape[:-1].numel()}")
    return inputs_embeds.masked_scatter(mask, image_features)


# =============================================================================================
# A8  tile merge                                       modeling_gar.py:248-260
# =============================================================================================
def merge_tiles(tiles: torch.Tensor, ncw: int, nch: int) -> torch.Tensor:
    b, n, c, th, tw = tiles.shape
    assert n == ncw * nch, f"{ncw * nch} != {n}"
    t = tiles.view(b, nch, ncw, c, th, tw).permute(0, 3, 1, 4, 2, 5).contiguous()
    return t.view(b, c, nch * th, ncw * tw)


# =============================================================================================
# A10  torchvision.ops.roi_align (CPU kernel), fp32    [EXT]  SURVEY.md A.3
# =============================================================================================
def roi_align(inp: torch.Tensor, rois: torch.Tensor, output_size, spatial_scale: float,
              sampling_ratio: int, aligned: bool) -> torch.Tensor:
    """inp [N,C,H,W] fp32, rois [K,5] fp32 (batch_idx,x1,y1,x2,y2) -> [K,C,ph,pw] fp32.
    Every scalar op is rounded to float32 in the order torchvision's roi_align_kernel.cpp evaluates it."""
    f32 = np.float32
    x = inp.detach().to(torch.float32).numpy()
    r = rois.detach().to(torch.float32).numpy()
    _, C, H, Wd = x.shape
    ph_n, pw_n = output_size
    out = np.zeros((r.shape[0], C, ph_n, pw_n), dtype=np.float32)
    ss = f32(spatial_scale)
    off = f32(0.5) if aligned else f32(0.0)
    for n in range(r.shape[0]):
        bi = int(r[n, 0])
        fm = x[bi]                                                          # [C,H,W]
        sw = f32(r[n, 1] * ss) - off
        sh = f32(r[n, 2] * ss) - off
        ew = f32(r[n, 3] * ss) - off
        eh = f32(r[n, 4] * ss) - off
        rw = f32(ew - sw)
        rh = f32(eh - sh)
        if not aligned:
            rw = max(rw, f32(1.0))
            rh = max(rh, f32(1.0))
        bh = f32(rh / f32(ph_n))
        bw = f32(rw / f32(pw_n))
        gh = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rh) / ph_n))
        gw = sampling_ratio if sampling_ratio > 0 else int(math.ceil(float(rw) / pw_n))
        count = f32(max(gh * gw, 1))
        for ph in range(ph_n):
            for pw in range(pw_n):
                acc = np.zeros((C,), dtype=np.float32)
                for iy in range(gh):
                    yy = f32(f32(sh + f32(f32(ph) * bh)) + f32(f32(f32(iy) + f32(0.5)) * bh) / f32(gh))
                    for ix in range(gw):
                        xx = f32(f32(sw + f32(f32(pw) * bw)) + f32(f32(f32(ix) + f32(0.5)) * bw) / f32(gw))
                        acc = acc + _bilinear(fm, H, Wd, yy, xx)
                out[n, :, ph, pw] = acc / count
    return torch.from_numpy(out)


def _bilinear(fm: np.ndarray, H: int, Wd: int, y, x) -> np.ndarray:
    f32 = np.float32
    if y < -1.0 or y > H or x < -1.0 or x > Wd:
        return np.zeros((fm.shape[0],), dtype=np.float32)
    if y <= 0:
        y = f32(0)
    if x <= 0:
        x = f32(0)
    yl = int(y)
    xl = int(x)
    if yl >= H - 1:
        yh = yl = H - 1
        y = f32(yl)
    else:
        yh = yl + 1
    if xl >= Wd - 1:
        xh = xl = Wd - 1
        x = f32(xl)
    else:
        xh = xl + 1
    ly = f32(y - f32(yl))
    lx = f32(x - f32(xl))
    hy = f32(f32(1.0) - ly)
    hx = f32(f32(1.0) - lx)
    w1, w2, w3, w4 = f32(hy * hx), f32(hy * lx), f32(ly * hx), f32(ly * lx)
    # torchvision: output_val += w1*v1 + w2*v2 + w3*v3 + w4*v4 (left to right, fp32)
    return ((w1 * fm[:, yl, xl] + w2 * fm[:, yl, xh]) + w3 * fm[:, yh, xl]) + w4 * fm[:, yh, xh]


# =============================================================================================
# A8-A11  RoI-aligned feature replay                   modeling_gar.py:348-414
# =============================================================================================
def replay_roi(bbox: Sequence[float], feat_h: int, feat_w: int, feat_stride: int):
    """Box math of modeling_gar.py:366-387 in Python floats (float64), incl. the reference's quirk that y
    uses the x-derived spatial_scale and that roi_align applies spatial_scale a second time.
    Returns (roi[5] as float32 list, spatial_scale)."""
    x1, y1, x2, y2 = [float(b) for b in bbox]
    orig_h, orig_w = feat_h * feat_stride, feat_w * feat_stride
    roi_orig_x1 = x1 * orig_w
    roi_orig_y1 = y1 * orig_h
    roi_orig_x2 = x2 * orig_w
    roi_orig_y2 = y2 * orig_h
    spatial_scale = feat_w / orig_w
    roi = [0.0, roi_orig_x1 * spatial_scale, roi_orig_y1 * spatial_scale,
           roi_orig_x2 * spatial_scale, roi_orig_y2 * spatial_scale]
    return roi, spatial_scale


def feature_replay(inputs_embeds: torch.Tensor, input_ids: torch.Tensor, image_features: torch.Tensor,
                   aspect_ratios, bboxes: List[dict], cfg, return_rois: bool = False):
    """The replay loop of GARModel.generate (modeling_gar.py:348-414). ``image_features`` [T+1, P*P, C];
    tile 0 (thumbnail) is dropped (:351)."""
    P = cfg.pooled_side
    tiles = image_features[1:].unsqueeze(0)                                  # b n (h w) c
    b, n, hw, c = tiles.shape
    tiles = tiles.reshape(b, n, P, P, c).permute(0, 1, 4, 2, 3)              # b n c h w
    new_embeds = []
    rois_dbg = []
    for bi in range(inputs_embeds.shape[0]):
        cur = inputs_embeds[bi]
        for crop_token in cfg.crop_tokens_ids:
            if crop_token in input_ids[bi]:
                idx = input_ids[bi].eq(crop_token).nonzero().squeeze()
                head_idx = int(idx.min())
                tail_idx = int(idx.max())
                ncw, nch = int(aspect_ratios[bi][0]), int(aspect_ratios[bi][1])
                fmap = merge_tiles(tiles, ncw, nch)
                feat_h, feat_w = fmap.shape[2:]
                roi, ss = replay_roi(bboxes[bi][str(crop_token)], feat_h, feat_w, cfg.feat_stride)
                roi_t = torch.tensor(roi, dtype=torch.float32)
                rf = roi_align(fmap.float(), roi_t.unsqueeze(0), (P, P), ss, 2, True)
                replay = rf.permute(0, 2, 3, 1).flatten(1, 2).to(fmap.dtype).squeeze(0)
                cur = torch.cat([cur[:head_idx], replay, cur[tail_idx + 1:]])
                rois_dbg.append((crop_token, head_idx, tail_idx, roi, ss))
        new_embeds.append(cur.unsqueeze(0))
    out = torch.cat(new_embeds, dim=0)
    return (out, rois_dbg) if return_rois else out


def feature_replay_video(inputs_embeds: torch.Tensor, input_ids: torch.Tensor, image_features: torch.Tensor,
                         bboxes: List[dict], frame_crop_token_ids: Sequence[int], cfg):
    """A13 — the video replay of PerceptionLMForConditionalGeneration.prepare_inputs_for_generation
    (modeling_perception_lm.py:765-852): ``image_features`` [F, P*P, C] is one pooled map per FRAME (no thumbnail
    drop, :772-774); frame f uses crop token ``<|reserved_special_token_{2+f}|>`` (:777-780) and RoI-aligns on its own
    P x P map (feat_h = feat_w = P, :787-816)."""
    P = cfg.pooled_side
    F_ = image_features.shape[0]
    tiles = image_features.unsqueeze(0).reshape(1, F_, P, P, -1).permute(0, 1, 4, 2, 3)     # b n c h w
    out = []
    for bi in range(inputs_embeds.shape[0]):
        cur = inputs_embeds[bi]
        for f in range(F_):
            crop_token = int(frame_crop_token_ids[f])
            if crop_token in input_ids[bi]:
                idx = input_ids[bi].eq(crop_token).nonzero().squeeze()
                head_idx, tail_idx = int(idx.min()), int(idx.max())
                roi, ss = replay_roi(bboxes[bi][str(crop_token)], P, P, cfg.feat_stride)
                rf = roi_align(tiles[:, f].float(), torch.tensor([roi], dtype=torch.float32), (P, P), ss, 2, True)
                replay = rf.permute(0, 2, 3, 1).flatten(1, 2).to(tiles.dtype).squeeze(0)
                cur = torch.cat([cur[:head_idx], replay, cur[tail_idx + 1:]])
        out.append(cur.unsqueeze(0))
    return torch.cat(out, dim=0)


# =============================================================================================
# A12  Llama (HF LlamaModel) + greedy loop            [EXT] transformers; SURVEY.md A.4
# =============================================================================================
def llama_inv_freq(tcfg) -> torch.Tensor:
    """transformers.modeling_rope_utils: default + rope_type 'llama3' scaling."""
    dim = tcfg.head_dim
    inv_freq = 1.0 / (tcfg.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float32) / dim))
    sc = tcfg.rope_scaling
    if sc and sc.get("rope_type", sc.get("type", "llama3")) == "llama3":
        factor = sc["factor"]
        low = sc["low_freq_factor"]
        high = sc["high_freq_factor"]
        old = sc["original_max_position_embeddings"]
        low_wl = old / low
        high_wl = old / high
        wavelen = 2 * math.pi / inv_freq
        inv_l = torch.where(wavelen > low_wl, inv_freq / factor, inv_freq)
        smooth = (old / wavelen - low) / (high - low)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        is_med = ~(wavelen < high_wl) * ~(wavelen > low_wl)
        inv_freq = torch.where(is_med, smoothed, inv_l)
    return inv_freq


def llama_rope_tables(tcfg, positions: torch.Tensor):
    inv = llama_inv_freq(tcfg)
    freqs = positions.to(torch.float32)[:, None] * inv[None, :]
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos(), emb.sin()                                              # [S, hd]


def _rotate_half(x):
    x1 = x[..., : x.shape[-1] // 2]
    x2 = x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def _rmsnorm(x, w, eps):
    var = x.pow(2).mean(-1, keepdim=True)
    return w * (x * torch.rsqrt(var + eps))


class KVCache:
    def __init__(self, n_layers):
        self.k = [None] * n_layers
        self.v = [None] * n_layers

    def update(self, i, k, v):
        if self.k[i] is None:
            self.k[i], self.v[i] = k, v
        else:
            self.k[i] = torch.cat([self.k[i], k], dim=2)
            self.v[i] = torch.cat([self.v[i], v], dim=2)
        return self.k[i], self.v[i]

    @property
    def length(self):
        return 0 if self.k[0] is None else self.k[0].shape[2]


def llama_forward(inputs_embeds: torch.Tensor, W: Dict[str, torch.Tensor], tcfg, cache: KVCache,
                  attn_impl: str = "eager", return_layers: bool = False, attention_mask: torch.Tensor = None):
    """LlamaModel.forward over ``inputs_embeds`` [B,S,C] appended after ``cache.length`` positions.
    Returns final-normed hidden states [B,S,C].

    ``attention_mask`` [B, cache.length + S] (1 = real token, 0 = padding; HF generation left-pads): what
    GenerationMixin does with it for a Llama (transformers generation/utils.py prepare_inputs_for_generation +
    masking_utils): position_ids = cumsum(mask) - 1 with padding positions set to 1, and a key is visible to a query iff
    it is causal AND not padding. Rows at padding positions are computed like any other (their values are never read by a
    real row)."""
    B, S, C = inputs_embeds.shape
    Hq, Hkv, hd = tcfg.num_attention_heads, tcfg.num_key_value_heads, tcfg.head_dim
    p0 = cache.length
    padded = attention_mask is not None and not bool((attention_mask != 0).all())
    if padded:
        am = (attention_mask != 0)
        assert am.shape == (B, p0 + S)
        pid = am.long().cumsum(-1) - 1
        pid = pid.masked_fill(~am, 1)[:, p0:]                                    # [B, S]
        inv = llama_inv_freq(tcfg)
        freqs = pid.to(torch.float32)[:, :, None] * inv[None, None, :]
        embp = torch.cat((freqs, freqs), dim=-1)
        cos, sin = embp.cos()[:, None], embp.sin()[:, None]                      # [B, 1, S, hd]
        attn_impl = "eager"
    else:
        pos = torch.arange(p0, p0 + S)
        cos, sin = llama_rope_tables(tcfg, pos)
    h = inputs_embeds
    layers = []
    for i in range(tcfg.num_hidden_layers):
        b = f"{LM}layers.{i}."
        x = _rmsnorm(h, W[b + "input_layernorm.weight"], tcfg.rms_norm_eps)
        q = F.linear(x, W[b + "self_attn.q_proj.weight"]).view(B, S, Hq, hd).transpose(1, 2)
        k = F.linear(x, W[b + "self_attn.k_proj.weight"]).view(B, S, Hkv, hd).transpose(1, 2)
        v = F.linear(x, W[b + "self_attn.v_proj.weight"]).view(B, S, Hkv, hd).transpose(1, 2)
        q = q * cos + _rotate_half(q) * sin
        k = k * cos + _rotate_half(k) * sin
        k_all, v_all = cache.update(i, k, v)
        rep = Hq // Hkv
        kk = k_all.repeat_interleave(rep, dim=1)
        vv = v_all.repeat_interleave(rep, dim=1)
        L = kk.shape[2]
        if attn_impl == "sdpa":
            if S == L:
                a = F.scaled_dot_product_attention(q, kk, vv, is_causal=True)
            else:
                m = torch.ones(S, L, dtype=torch.bool).tril(diagonal=L - S)
                a = F.scaled_dot_product_attention(q, kk, vv, attn_mask=m)
        else:
            s = (q @ kk.transpose(2, 3)) * (hd ** -0.5)
            m = torch.ones(S, L, dtype=torch.bool).tril(diagonal=L - S)
            if padded:
                m = m[None, None] & am[:, None, None, :]                         # [B, 1, S, L]: causal and not padding
            s = s.masked_fill(~m, torch.finfo(s.dtype).min)
            a = torch.softmax(s, dim=-1, dtype=torch.float32) @ vv
        a = a.transpose(1, 2).reshape(B, S, Hq * hd)
        h = h + F.linear(a, W[b + "self_attn.o_proj.weight"])
        x = _rmsnorm(h, W[b + "post_attention_layernorm.weight"], tcfg.rms_norm_eps)
        g = F.linear(x, W[b + "mlp.gate_proj.weight"])
        u = F.linear(x, W[b + "mlp.up_proj.weight"])
        h = h + F.linear(F.silu(g) * u, W[b + "mlp.down_proj.weight"])
        if return_layers:
            layers.append(h.clone())
    out = _rmsnorm(h, W[LM + "norm.weight"], tcfg.rms_norm_eps)
    return (out, layers) if return_layers else out


def lm_head_weight(W, tcfg):
    return W["mllm.lm_head.weight"] if "mllm.lm_head.weight" in W else W[LM + "embed_tokens.weight"]


def greedy_generate(inputs_embeds: torch.Tensor, W, tcfg, max_new_tokens: int, eos_token_id=None,
                    attn_impl: str = "eager", return_logits: bool = False, attention_mask: torch.Tensor = None):
    """GenerationMixin greedy search started from ``inputs_embeds`` (modeling_gar.py:418-426): returns only
    the new tokens [B, n_new]. Stops when every sequence has produced EOS (finished rows emit EOS as pad).
    ``attention_mask`` [B, S] of a left-padded batch is extended by a column of ones per generated token."""
    B = inputs_embeds.shape[0]
    am = None if attention_mask is None else (attention_mask != 0).long()
    E = W[LM + "embed_tokens.weight"]
    head = lm_head_weight(W, tcfg)
    cache = KVCache(tcfg.num_hidden_layers)
    eos = set()
    if eos_token_id is not None:
        eos = set(eos_token_id) if isinstance(eos_token_id, (list, tuple)) else {int(eos_token_id)}
    h = llama_forward(inputs_embeds, W, tcfg, cache, attn_impl, attention_mask=am)
    out_tokens, all_logits = [], []
    finished = torch.zeros(B, dtype=torch.bool)
    pad = next(iter(eos)) if eos else 0
    for step in range(max_new_tokens):
        logits = F.linear(h[:, -1, :], head)                                 # [B, V]
        if return_logits:
            all_logits.append(logits.clone())
        nxt = torch.argmax(logits, dim=-1)
        nxt = torch.where(finished, torch.full_like(nxt, pad), nxt)
        out_tokens.append(nxt)
        if eos:
            finished = finished | torch.tensor([int(t) in eos for t in nxt])
            if bool(finished.all()):
                break
        if step + 1 < max_new_tokens:
            if am is not None:
                am = torch.cat([am, torch.ones(B, 1, dtype=am.dtype)], dim=1)
            h = llama_forward(F.embedding(nxt, E).unsqueeze(1), W, tcfg, cache, attn_impl, attention_mask=am)
    seq = torch.stack(out_tokens, dim=1)
    return (seq, torch.stack(all_logits, dim=1)) if return_logits else seq


# =============================================================================================
# GARModel.generate end to end                          modeling_gar.py:295-428
# =============================================================================================
def build_inputs_embeds(W, cfg, pixel_values, global_mask_values, aspect_ratios, bboxes, input_ids,
                        attn_impl: str = "eager", return_intermediates: bool = False, video_frame_tokens=None):
    pixel_values = pixel_values.to(torch.float32)
    binary = decode_mask_values(global_mask_values, cfg.prompt_numbers)
    mask_embeds = mask_patch_embed(binary, W["mask_patch_embedding.weight"])
    image_features = get_image_features(pixel_values, mask_embeds, W, cfg, attn_impl)
    inputs_embeds = embed_and_scatter(input_ids, W[LM + "embed_tokens.weight"], image_features,
                                      cfg.mllm_config.image_token_id)
    if video_frame_tokens is not None:
        replayed = feature_replay_video(inputs_embeds, input_ids, image_features, bboxes, video_frame_tokens, cfg)
    else:
        replayed = feature_replay(inputs_embeds, input_ids, image_features, aspect_ratios, bboxes, cfg)
    if return_intermediates:
        return replayed, {"mask_embeds": mask_embeds, "image_features": image_features,
                          "inputs_embeds_scattered": inputs_embeds}
    return replayed


def gar_generate(W, cfg, pixel_values, global_mask_values, aspect_ratios, bboxes, input_ids,
                 attention_mask=None, max_new_tokens: int = 64, eos_token_id=None,
                 attn_impl: str = "eager", return_logits: bool = False, video_frame_tokens=None):
    """Restates GARModel.generate for fp32 CPU tensors. ``attention_mask`` is forwarded to the Llama generate like the
    reference does (modeling_gar.py:418-426; its own callers pass all ones, eval_dataset.py:143; a left-padded batch
    follows HF's position / masking rules, see llama_forward). ``video_frame_tokens`` selects the video replay (A13)."""
    embeds = build_inputs_embeds(W, cfg, pixel_values, global_mask_values, aspect_ratios, bboxes, input_ids,
                                 attn_impl, video_frame_tokens=video_frame_tokens)
    return greedy_generate(embeds, W, cfg.mllm_config.text_config, max_new_tokens, eos_token_id, attn_impl,
                           return_logits, attention_mask=attention_mask)
