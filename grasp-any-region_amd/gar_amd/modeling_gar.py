"""GARModel — host-side sequencing of the region-captioning hot path on one MI355X.

Keeps the call surface of the reference's ``GARModel`` (projects/grasp_any_region/hf_models/modeling_gar.py):
``model.config.prompt_numbers``, ``model.generate(input_ids=, attention_mask=, pixel_values=, global_mask_values=,
bboxes=, aspect_ratios=, generation_config=, return_dict=)`` -> object with ``.sequences`` (new tokens only, :418-426),
``get_image_features`` (modeling_perception_lm.py:239-269).  Everything numeric is a HIP kernel behind the C ABI
(``gar_amd.ops``); this file only allocates buffers, orders launches and captures the decode step in a hipGraph.

Differences by design: the mask conv + patch-embed conv are one im2col GEMM; ``_merge`` and the fp32
copy of the feature map are never materialised (``gar_roi_replay`` indexes the tile layout); placeholder / crop-token
bookkeeping runs on the device, so ``generate`` has no host sync before the first EOS check.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence

import torch

from . import hip, ops
from .configuration_gar import GARConfig
from .planner import plan_chunks
from .weights import LM, PJ, VT, check_weights, load_weights, normalize_checkpoint, synthetic_weights

LOG2E = 1.4426950408889634


# 16-bit element types: the fused bf16 kernels serve both (libgar_hip.so / its twin libgar_hip_f16.so, hip.lib(dtype))
HALF_DTYPES = (torch.bfloat16, torch.float16)

class _PendingGeneration:
    """What generate_begin hands to generate_finish: the KV state (slot) the prompt was prefilled into and the loop's settings."""
    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.ready = None


@dataclass
class GenerateOutput:
    sequences: torch.Tensor                      # [B, n_new] int64 (only new tokens, as with inputs_embeds in HF)
    logits: Optional[torch.Tensor] = None        # [B, n_new, V] when return_logits=True (tests)
    # validate=False only: device int32 [1], OR of the INPUT_* bits below found by the device-side checks (0 = the
    # inputs were what validate=True would have accepted). Reading it is the caller's sync; generate() itself raises on it
    # wherever it synchronises anyway (EOS polls).
    input_flags: Optional[torch.Tensor] = None
    # GenerationPipeline only: event on the decode stream behind this output's last kernel — a consumer on another stream
    # (the caller's) waits for it before it reads ``sequences``
    done: Optional[object] = None


INPUT_COUNT_MISMATCH, INPUT_SPAN_LENGTH, INPUT_MISSING_BBOX, INPUT_ID_RANGE, INPUT_MASK_NOT_LEFT_PADDED = 1, 2, 4, 8, 16


def describe_input_flags(flags: int) -> str:
    names = {INPUT_COUNT_MISMATCH: "image token count != image feature rows (reference: ValueError)",
             INPUT_SPAN_LENGTH: "a crop-token span is not P*P long (the reference's splice would change the sequence length)",
             INPUT_MISSING_BBOX: "a crop token present in input_ids has no bbox (reference: KeyError)",
             INPUT_ID_RANGE: "input_ids outside [0, vocab)",
             INPUT_MASK_NOT_LEFT_PADDED: "attention_mask is not LEFT-padded (a row is not 0...01...1): HF generation would continue "
                                         "such a row after its padding"}
    return "; ".join(v for k, v in names.items() if flags & k)


# GenerationConfig fields the reference forwards to HF's generate (modeling_gar.py:418-426) that would change the tokens of a
# greedy search and are NOT implemented here: name -> the value(s) that leave greedy search unchanged. Anything else raises
# instead of being ignored (the reference's own callers pass max_new_tokens / do_sample=False / eos / pad only).
_NEUTRAL_GENERATION_OPTIONS = {
    "num_beams": (None, 1), "num_beam_groups": (None, 1), "penalty_alpha": (None, 0, 0.0), "repetition_penalty": (None, 1, 1.0),
    "encoder_repetition_penalty": (None, 1, 1.0), "length_penalty": (None, 1, 1.0), "no_repeat_ngram_size": (None, 0),
    "diversity_penalty": (None, 0, 0.0), "bad_words_ids": (None,), "force_words_ids": (None,), "constraints": (None,),
    "forced_bos_token_id": (None,), "forced_eos_token_id": (None,), "suppress_tokens": (None,), "begin_suppress_tokens": (None,),
    "sequence_bias": (None,), "min_length": (None, 0), "min_new_tokens": (None, 0), "num_return_sequences": (None, 1),
    "prompt_lookup_num_tokens": (None,), "assistant_model": (None,),
}
# sampling-only knobs: without do_sample they do not touch greedy search (HF warns and ignores them) — accepted
_SAMPLING_ONLY_OPTIONS = ("temperature", "top_k", "top_p", "min_p", "typical_p", "epsilon_cutoff", "eta_cutoff")


# keywords generate() accepts besides its own parameters: the generation options it reads or refuses by value
_GENERATION_KWARGS = set(_NEUTRAL_GENERATION_OPTIONS) | {"do_sample", "temperature", "top_k", "top_p", "min_p", "typical_p", "epsilon_cutoff",
                                                          "eta_cutoff", "max_new_tokens", "eos_token_id", "pad_token_id", "use_cache",
                                                          "return_dict_in_generate", "output_scores", "output_logits", "output_attentions",
                                                          "synced_gpus", "streamer", "labels"}


class _OverlaidOptions:
    """generation options: keyword arguments laid over a GenerationConfig / dict (HF: `generate(generation_config, **kwargs)`)"""

    def __init__(self, base_get, over):
        self._base, self._over = base_get, over

    def get(self, k, d=None):
        return self._over[k] if k in self._over else self._base(k, d)


def _refuse_non_greedy(get):
    for name, neutral in _NEUTRAL_GENERATION_OPTIONS.items():
        v = get(name)
        if not any((v is n) or (n is not None and not isinstance(v, bool) and v == n) for n in neutral):
            raise hip.GarError(f"generation option {name}={v!r} is not implemented (greedy search only: it would change the tokens "
                               f"and is refused rather than ignored)")


def _sampling_options(get):
    """(temperature, top_p, top_k) of a do_sample = True request, with HF's defaults (GenerationConfig: temperature 1.0, top_k 50,
    top_p 1.0) and HF's argument checks (TemperatureLogitsWarper / TopKLogitsWarper / TopPLogitsWarper). The warpers that are not
    built (min_p, typical_p, epsilon / eta cutoff) are refused when they would act."""
    for name, neutral in (("min_p", (None, 0, 0.0)), ("typical_p", (None, 1, 1.0)), ("epsilon_cutoff", (None, 0, 0.0)),
                          ("eta_cutoff", (None, 0, 0.0))):
        v = get(name)
        if not any((v is n) or (n is not None and not isinstance(v, bool) and v == n) for n in neutral):
            raise hip.GarError(f"sampling option {name}={v!r} is not implemented (temperature, top_k and top_p are)")
    temperature = get("temperature", 1.0)
    temperature = 1.0 if temperature is None else float(temperature)
    if not temperature > 0:
        raise ValueError(f"`temperature` (={temperature}) has to be a strictly positive float")
    top_k = get("top_k", 50)
    top_k = 0 if top_k is None else int(top_k)
    if top_k < 0:
        raise ValueError(f"`top_k` has to be a non-negative integer (0 = off), but is {top_k}")
    top_p = get("top_p", 1.0)
    top_p = 1.0 if top_p is None else float(top_p)
    if not 0 < top_p <= 1.0:
        raise ValueError(f"`top_p` has to be a float > 0 and <= 1, but is {top_p}")
    return temperature, top_p, top_k


def _round_up(x, m):
    return (x + m - 1) // m * m


def _rope2d_tables(v) -> (torch.Tensor, torch.Tensor):
    """timm RotaryEmbeddingCat tables: host-side constant generation, fp32."""
    hd, g, nb = v.head_dim, v.grid, v.head_dim // 4
    bands = 1.0 / (v.rope_temperature ** (torch.arange(0, nb, dtype=torch.int64).to(torch.float32) / nb))
    t = torch.arange(g, dtype=torch.int64).to(torch.float32) + v.rope_grid_offset
    g0, g1 = torch.meshgrid(t, t, indexing=v.rope_grid_indexing)
    pos = torch.stack([g0, g1], dim=-1).unsqueeze(-1) * bands
    sin = pos.sin().reshape(g * g, -1).repeat_interleave(2, -1)
    cos = pos.cos().reshape(g * g, -1).repeat_interleave(2, -1)
    return sin.contiguous(), cos.contiguous()


def _llama_inv_freq(t) -> torch.Tensor:
    """HF rope init incl. rope_type 'llama3' (SURVEY.md A.4): host-side constant generation, fp32."""
    dim = t.head_dim
    inv = 1.0 / (t.rope_theta ** (torch.arange(0, dim, 2, dtype=torch.int64).to(torch.float32) / dim))
    sc = t.rope_scaling
    if sc and sc.get("rope_type", sc.get("type", "llama3")) == "llama3":
        factor, low, high, old = sc["factor"], sc["low_freq_factor"], sc["high_freq_factor"], \
            sc["original_max_position_embeddings"]
        wavelen = 2 * math.pi / inv
        inv_l = torch.where(wavelen > old / low, inv / factor, inv)
        smooth = (old / wavelen - low) / (high - low)
        smoothed = (1 - smooth) * inv_l / factor + smooth * inv_l
        med = ~(wavelen < old / high) * ~(wavelen > old / low)
        inv = torch.where(med, smoothed, inv_l)
    return inv


def _on_model_device(fn):
    """Run a public entry point with the model's device current: the C ABI launches on the current HIP device and
    ``hip.stream()`` hands it torch's current stream of THAT device, so a model on cuda:1 must not enqueue on device 0."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)
    return wrapped


class _InputEmbeddings:
    """callable returned by ``get_input_embeddings()``: ids [...] -> rows of embed_tokens [..., C] (gar_embed_lookup)."""

    def __init__(self, model):
        self._m = model
        self.weight = model.E

    def __call__(self, input_ids: torch.Tensor) -> torch.Tensor:
        m = self._m
        with torch.cuda.device(m.device):
            ids = input_ids.to(m.device, torch.int64).contiguous()
            out = torch.empty(*ids.shape, m.E.shape[1], dtype=m.dtype, device=m.device)
            if ids.numel():
                ops.embed_lookup(ids.view(-1), m.E, out.view(-1, m.E.shape[1]))
        return out


class _MllmFacade:
    """``model.mllm`` of the reference's GARModel (modeling_gar.py:47-50): the helper calls its generate() makes
    (:332-346) for callers that script the stages themselves. Each method is the HIP kernel ``generate`` itself uses
    (or a pure index view); the reference's own callers in scope never touch them."""

    def __init__(self, model):
        self._m = model
        self.config = model.config.mllm_config

    @property
    def dtype(self):
        return self._m.dtype

    @property
    def device(self):
        return self._m.device

    def get_input_embeddings(self):
        return _InputEmbeddings(self._m)

    def get_image_features(self, pixel_values, mask_embeds=None, global_mask_values=None, **kw):
        """modeling_perception_lm.py:239-269, the reference's signature: ``mask_embeds`` [T, C_v, g, g] is the output of
        ``model.mask_patch_embedding`` (modeling_gar.py:326-337) and is added to the patch embeddings
        (modeling_perception_lm.py:195-196). ``generate`` does not come through here: it hands the processor's
        ``global_mask_values`` to the patch-embed GEMM, which carries the mask convolution as extra K columns; that form is
        available to callers too (``global_mask_values=``; not both)."""
        if mask_embeds is not None and global_mask_values is not None:
            raise hip.GarError("pass either mask_embeds (the reference's form) or global_mask_values (the fused form), not both")
        return self._m.get_image_features(pixel_values, global_mask_values, mask_embeds=mask_embeds)

    def get_placeholder_mask(self, input_ids, inputs_embeds, image_features=None, video_features=None):
        """(special_image_mask, special_video_mask) expanded to ``inputs_embeds``' shape, with the reference's count check
        (modeling_perception_lm.py:271-331) — from gar_placeholder_scan's slot table."""
        m = self._m
        if input_ids is None:
            raise hip.GarError("get_placeholder_mask needs input_ids (the embeds-only form of the reference is not built)")
        with torch.cuda.device(m.device):
            ids = input_ids.to(m.device, torch.int64).contiguous()
            B, S = ids.shape
            masks = []
            for tok, feats, what in ((self.config.image_token_id, image_features, "Image"),
                                     (self.config.video_token_id, video_features, "Videos")):
                slot = torch.empty(B, S, dtype=torch.int32, device=m.device)
                counts = torch.empty(B, dtype=torch.int32, device=m.device)
                spans = torch.empty(B, 1, 2, dtype=torch.int32, device=m.device)
                ops.placeholder_scan(ids, tok, m.crop_ids_dev[:1], slot, counts, spans)
                n_tok = int(counts.sum().item())
                if feats is not None and n_tok * inputs_embeds.shape[-1] != feats.numel():
                    raise ValueError(f"{what} features and image tokens do not match: tokens: {n_tok}, features "
                                     f"{feats.numel() // feats.shape[-1]}")
                masks.append((slot >= 0).unsqueeze(-1).expand_as(inputs_embeds))
        return masks[0], masks[1]


class _MaskPatchEmbedding:
    """``model.mask_patch_embedding`` of the reference (nn.Conv2d(3, C_v, 14, 14, bias=False), modeling_gar.py:54-60) as a
    callable: binary mask tiles [T, 3, H, W] -> mask embeddings [T, C_v, g, g] (modeling_gar.py:326-328). The mask half of the
    patch-embed GEMM run on its own (gar_patch_embed with zero weights in the pixel slots, or gar_patch_im2col + gar_gemm); the
    result is a channel-first VIEW of token-major storage, which is what ``mllm.get_image_features(mask_embeds=)`` reads."""

    def __init__(self, model):
        self._m = model

    @property
    def weight(self):       # [C_v, 3, p, p], reconstructed from the fused patch-embed operand (fp32 master in model dtype)
        m = self._m
        v = m.config.mllm_config.vision_config
        pp = v.patch_size * v.patch_size
        return m.w_patch[:, 3 * pp:6 * pp].reshape(-1, 3, v.patch_size, v.patch_size)

    def __call__(self, binary_mask: torch.Tensor) -> torch.Tensor:
        return self._m._mask_embed(binary_mask)


class GenerationPipeline:
    """Software pipeline over consecutive ``generate`` calls: the decode loop of batch i on a second HIP stream, the prompt
    phase (vision tower, sequence assembly, prefill) of batch i + 1 on the caller's stream, two KV-state slots. The decode step
    is HBM-bound (weights and KV cache stream, matrix pipes idle), the prompt phase MFMA-bound: wherever the hardware can
    co-schedule their workgroups — under the attention kernels and at kernel boundaries; the persistent tile GEMM owns its
    CUs — the decode loop costs no wall time. Same kernels on the same data as ``model.generate``: bit-identical outputs.

    ``submit(sample)`` returns the outputs that became available (at most one, of the previous batch; stream-ordered, not
    host-synchronised), ``flush()`` the last one and makes the caller's stream wait for the decode stream."""

    PRIORITY = 0        # of the decode stream (-1 = high)

    def __init__(self, model: "GARModel", **generate_kwargs):
        self.model, self.kwargs = model, generate_kwargs
        self.side = None
        self.done = {}              # slot -> event: that slot's previous decode loop has finished
        self.prev = None
        self.n = 0

    def _finish(self, pend):
        m = self.model
        if self.side is None:
            with torch.cuda.device(m.device):
                self.side = torch.cuda.Stream(device=m.device, priority=self.PRIORITY)
        self.side.wait_event(pend.ready)
        with torch.cuda.device(m.device), torch.cuda.stream(self.side):
            out = m.generate_finish(pend)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.done[pend.st["slot"]] = out.done = ev
        return out

    def submit(self, sample) -> List[GenerateOutput]:
        m = self.model
        outs = []
        # Without EOS polling the decode loop enqueues without a host sync: put it in flight FIRST, so that nothing the prompt
        # phase's host code waits for can delay it. With EOS polling generate_finish blocks the host until its batch is done:
        # then the next prompt phase has to be enqueued before it.
        early = self.prev is not None and not self.prev.eos
        if early:
            outs.append(self._finish(self.prev))
            self.prev = None
        slot = self.n & 1
        self.n += 1
        with torch.cuda.device(m.device):
            main = torch.cuda.current_stream(m.device)
            if slot in self.done:
                main.wait_event(self.done[slot])            # the prefill overwrites that slot's cache, counters and token buffer
            pend = m.generate_begin(**sample, **self.kwargs, state_slot=slot)
            pend.ready = torch.cuda.Event()
            pend.ready.record(main)
        if self.prev is not None:
            outs.append(self._finish(self.prev))
        self.prev = pend
        return outs

    def flush(self) -> List[GenerateOutput]:
        outs = []
        if self.prev is not None:
            outs.append(self._finish(self.prev))
            self.prev = None
        if self.side is not None:
            with torch.cuda.device(self.model.device):
                torch.cuda.current_stream(self.model.device).wait_stream(self.side)
        return outs


class GARModel:
    # bf16: the LayerNorms of the ViT blocks and the RMSNorms of the Llama prefill are folded into the GEMM pairs around
    # them (gar_gemm_params.row_scale / row_stats): the consumer GEMM reads the residual stream itself with a weight that
    # carries gamma (and, for LayerNorm, centred rows), the producer GEMM's epilogue writes the row statistics — the
    # stand-alone norm passes (3.3 % of a step, at HBM bandwidth) disappear
    FOLD_NORMS = True
    # bf16 with FOLD_NORMS: RoPE, q scale and the KV-cache append run in the Llama prefill qkv GEMM's epilogue
    # (GAR_EPI_QKV_ROPE_LLM) — no [B*S, (Hq + 2 Hkv) hd] intermediate, no llm_qkv_post pass
    LLM_QKV_EPILOGUE = True
    DECODE_GU_NORM_FOLDED = True  # bf16, 16 < B <= 64: post-attention RMSNorm inside the gate/up GEMM (gar_gemm_params.norm_folded)
    DECODE_ATTN_TAKES_QKV = True  # bf16: the decode attention launch does llm_qkv_post's work (gar_attention_decode_qkv)
    VIT_CLS_KEY_FOLD = True       # bf16: the cls key / value row enters the ViT attention through the initial softmax state
    VIT_V_ROW_MAJOR = True        # bf16: v leaves the qkv GEMM head-major, gar_attention_vrow transposes on its LDS reads
    # split-KV decode attention: kv splits per (sequence, kv head) so that ~DECODE_ATTN_BLOCKS workgroups exist, at most
    # DECODE_ATTN_MAX_SPLITS — measured optimum at kv ~ 4.75k (tools/bench_decode_attn.py, hipGraph replays, round 4): B = 1 / 2 / 4:
    # 8 splits (12.0 us against 21.9 with the 64 splits a 512-block target gave: more splits = a longer combine launch and more
    # partial-block overhead than bytes in flight are worth), B = 8: 2-4, B = 12: 2, B >= 16: 1 (no combine launch)
    DECODE_ATTN_BLOCKS = 192
    DECODE_ATTN_MAX_SPLITS = 8
    FUSE_NORM_MAX_BATCH = 16      # largest decode batch that keeps `down` un-split (f32 / plain weights: RMSNorm in the GEMV prologue)
    # The LAST Llama layer of a prefill computes attention / o / gate-up / down for the last prompt row of every sequence only:
    # the head reads nothing else (modeling_perception_lm.py:545-552, `lm_head(hidden_states[:, slice_indices, :])`) and the
    # decode steps read the layer's K / V rows, which the qkv GEMM still writes for every row. Exact w.r.t. the reference,
    # which computes and discards the other S - 1 rows; 2.5 % of a region's GEMM + attention work at GAR-1B / 1024^2.
    PRUNE_LAST_PREFILL_LAYER = True

    def __init__(self, config: GARConfig, weights: Dict[str, torch.Tensor], dtype: torch.dtype = torch.bfloat16,
                 device: str = "cuda:0", prefill_chunk: Optional[int] = None, keep_plain_weights: bool = False):
        self.device = torch.device(device)
        # bf16 keeps ONE copy of every weight: the norm-folded forms (ViT qkv / fc1, Llama qkv / gate-up) serve the tile GEMMs
        # (row_scale), the decode GEMVs (norm_folded) and — behind a unit-gain norm pass — the shapes neither takes.
        # keep_plain_weights=True also keeps the un-folded copies (+1.6 GB GAR-1B, +11 GB GAR-8B): A/B switch for FOLD_NORMS = False
        self.keep_plain_weights = bool(keep_plain_weights)
        hip.require_device(self.device.index or 0)
        self.config = config
        self.dtype = dtype
        self.prompt_numbers = config.prompt_numbers
        self.crop_tokens_ids = list(config.crop_tokens_ids)
        check_weights(config, weights)
        with torch.cuda.device(self.device):
            self._prepare_weights(weights)
            self._pack_weights()
        self.mllm = _MllmFacade(self)
        self.mask_patch_embedding = _MaskPatchEmbedding(self)
        self._mask_only_w = {}
        self._ws: Dict[tuple, Dict[str, torch.Tensor]] = {}
        self._graphs: Dict[tuple, object] = {}
        self._llm_lru: List[tuple] = []
        self._video_crop_ids: Dict[tuple, torch.Tensor] = {}
        # None: the vision tower runs over chunks of image TILES and the prefill over chunks of SEQUENCES, both chosen by
        # gar_amd.planner so that the persistent tile GEMMs run whole rounds (decode serves all B at once).
        # An int pins both passes to chunks of that many regions (the pre-planner behaviour; bench.py --prefill-chunk).
        self.prefill_chunk = None if prefill_chunk is None else int(prefill_chunk)

    # ---- construction -------------------------------------------------------------------------------------------
    @classmethod
    def from_synthetic(cls, config: GARConfig, seed: int = 0, dtype=torch.bfloat16, device="cuda:0", **kw):
        return cls(config, synthetic_weights(config, seed), dtype, device, **kw)

    @classmethod
    def from_pretrained(cls, path: str, dtype=torch.bfloat16, device="cuda:0", config: GARConfig = None):
        import os
        if config is None:
            config = GARConfig.from_json_file(os.path.join(path, "config.json"))
        return cls(config, normalize_checkpoint(config, load_weights(path)), dtype, device)

    @classmethod
    def from_shapes(cls, config: GARConfig, dtype=torch.bfloat16, device="cuda:0"):
        """Replica with UNINITIALISED device weights of the right shapes — to be filled by ``broadcast_weights``
        (non-source ranks of the data-parallel runner never touch host weights)."""
        from .weights import weight_shapes
        W = {k: torch.empty(s, dtype=dtype, device=device) for k, s in weight_shapes(config).items()}
        return cls(config, W, dtype, device)

    def _weight_slots(self):
        """where every prepared weight tensor lives: (container, key, index-in-tuple or None), in a fixed order"""
        own = self.__dict__
        slots = [(own, "w_patch", None), (own, "pos", None), (own, "norm_pre", 0), (own, "norm_pre", 1)]
        slots += [(self.pj, k, None) for k in self.pj]
        slots += [(own, "E", None), (own, "final_norm", None)]
        if self.cls is not None:
            slots.append((own, "cls", None))
        if self.w_patch_gather is not None:
            slots.append((own, "w_patch_gather", None))
        if self.lm_head is not self.E:
            slots.append((own, "lm_head", None))
        for d in list(self.vblocks) + list(self.layers):
            for k, v in d.items():
                slots += [(d, k, i) for i in range(len(v))] if isinstance(v, tuple) else [(d, k, None)]
        return slots

    def weight_tensors(self) -> List[torch.Tensor]:
        return [c[k] if i is None else c[k][i] for c, k, i in self._weight_slots()]

    def _pack_weights(self):
        """Move every prepared weight into ONE allocation per dtype (``self.arenas``) and re-bind the tensors as views of it: the
        data-parallel weight exchange is then one in-place RCCL broadcast per arena — no concatenated bucket
        copies — and a replica's weights are one contiguous range of HBM."""
        from .weights import pack_arenas
        tied = self.lm_head is self.E
        slots = self._weight_slots()
        self.arenas, views = pack_arenas([c[k] if i is None else c[k][i] for c, k, i in slots])
        for (c, k, i), v in zip(slots, views):
            if i is None:
                c[k] = v
            else:
                c[k] = tuple(v if j == i else x for j, x in enumerate(c[k]))
        if tied:
            self.lm_head = self.E

    @_on_model_device
    def broadcast_weights(self, src: int = 0):
        """RCCL broadcast of the prepared weights from rank ``src``: one collective per dtype arena, in place (one-off)."""
        from .dp import broadcast_arenas
        n = broadcast_arenas(self.arenas, src)
        self._mask_only_w.clear()       # derived from w_patch / w_patch_gather (mask_patch_embedding): rebuilt from the new weights
        return n

    def weight_arena_bytes(self) -> int:
        return sum(a.numel() * a.element_size() for a in self.arenas.values())

    def eval(self):
        return self

    def get_input_embeddings(self):
        """modeling_gar.py:67-68"""
        return _InputEmbeddings(self)

    @staticmethod
    def _merge(tiles: torch.Tensor, ncw: int, nch: int) -> torch.Tensor:
        """modeling_gar.py:248-260: [B, nch*ncw, C, th, tw] tiles -> [B, C, nch*th, ncw*tw] image. A pure index permutation;
        ``generate`` never materialises it (gar_roi_replay indexes the tile layout), kept for callers that want the map."""
        b, n, c, th, tw = tiles.shape
        assert n == ncw * nch, f"{ncw * nch} != {n}"
        return tiles.view(b, nch, ncw, c, th, tw).permute(0, 3, 1, 4, 2, 5).contiguous().view(b, c, nch * th, ncw * tw)

    def _dev(self, t: torch.Tensor) -> torch.Tensor:
        return t.to(device=self.device, dtype=self.dtype).contiguous()

    def _prepare_weights(self, W: Dict[str, torch.Tensor]):
        cfg = self.config
        v, t = cfg.mllm_config.vision_config, cfg.mllm_config.text_config
        pp = v.patch_size * v.patch_size
        self.Kp = _round_up(6 * pp, 64)
        D = v.embed_dim
        wcat = torch.zeros(D, self.Kp, dtype=torch.float32, device=W[VT + "patch_embed.proj.weight"].device)
        wcat[:, :3 * pp] = W[VT + "patch_embed.proj.weight"].float().flatten(1)
        wcat[:, 3 * pp:6 * pp] = W["mask_patch_embedding.weight"].float().flatten(1)
        d = self._dev
        self.w_patch = d(wcat)
        # the same two conv weights in the K order of gar_patch_embed (patches gathered from the image tiles by the GEMM's
        # DMA): column ((tensor*3 + c)*4 + ky//4)*64 + (ky%4)*16 + kx; zero in the slots no pixel of the patch maps to
        self.w_patch_gather = None
        P_ = v.patch_size
        Kg = ops.patch_embed_k(v.img_size, P_) if self.dtype in HALF_DTYPES else 0
        if Kg:
            wg = torch.zeros(D, 2, 3, 4, 4, 16, dtype=torch.float32, device=wcat.device)
            for ti, key in enumerate((VT + "patch_embed.proj.weight", "mask_patch_embedding.weight")):
                w4 = torch.zeros(D, 3, 16, 16, dtype=torch.float32, device=wcat.device)
                w4[:, :, :P_, :P_] = W[key].float()
                wg[:, ti] = w4.view(D, 3, 4, 4, 16)
            self.w_patch_gather = d(wg.view(D, Kg))
        self.npt = 1 if cfg.mllm_config.vision_use_cls_token else 0
        self.cls = d(W[VT + "cls_token"].reshape(-1)) if self.npt else None
        self.pos = d(W[VT + "pos_embed"].reshape(-1, D))
        self.norm_pre = (d(W[VT + "norm_pre.weight"]), d(W[VT + "norm_pre.bias"]))
        self.vblocks = []
        # The vision attention path is built for head_dim 64, 128 and — bf16 only — 96 (PE-G/14, GAR-8B). Any other head
        # dim (and 96 in the f32 parity mode) runs zero-padded to the next built size: padded q/k/v rows of the fused qkv
        # weight and bias and padded input columns of the output projection are zero (and the RoPE table rotates them by
        # the identity), so scores and outputs are unchanged.
        H, hd = v.num_heads, v.head_dim
        if hd > 128:
            raise hip.GarError(f"vision head_dim {hd} > 128 is not built")
        native = (64, 96, 128) if self.dtype in HALF_DTYPES else (64, 128)
        self.v_hd = hdp = hd if hd in native else (64 if hd < 64 else 128)

        def pad_qkv(w):          # [3*H*hd, ...] -> [3*H*hdp, ...]
            if hdp == hd:
                return w
            w = w.reshape(3, H, hd, *w.shape[1:])
            out = w.new_zeros(3, H, hdp, *w.shape[3:])
            out[:, :, :hd] = w
            return out.reshape(3 * H * hdp, *w.shape[3:])

        def pad_proj(w):         # [D, H*hd] -> [D, H*hdp]
            if hdp == hd:
                return w
            out = w.new_zeros(w.shape[0], H, hdp)
            out[:, :, :hd] = w.reshape(w.shape[0], H, hd)
            return out.reshape(w.shape[0], H * hdp)

        fold = self.dtype in HALF_DTYPES
        plain = not fold or self.keep_plain_weights

        def fold_ln(w, bias, gamma, beta):
            """LN(x) W^T + b = rstd * (x Wc^T) + b': Wc = W diag(gamma) with every row's mean removed (the mean subtraction
            of the LayerNorm is linear and is absorbed by the weight), b' = b + W beta. fp32 in, model dtype out."""
            wg = w.float() * gamma.float()[None, :]
            return d(wg - wg.mean(dim=1, keepdim=True)), d(bias.float() + w.float() @ beta.float())

        for i in range(v.depth):
            b = f"{VT}blocks.{i}."
            blk = {}
            if fold:
                qw, qb = pad_qkv(W[b + "attn.qkv.weight"]), pad_qkv(W[b + "attn.qkv.bias"])
                blk["qkv_wf"], blk["qkv_bf"] = fold_ln(qw, qb, W[b + "norm1.weight"], W[b + "norm1.bias"])
                blk["fc1_wf"], blk["fc1_bf"] = fold_ln(W[b + "mlp.fc1.weight"], W[b + "mlp.fc1.bias"],
                                                       W[b + "norm2.weight"], W[b + "norm2.bias"])
            if plain:
                blk.update(n1=(d(W[b + "norm1.weight"]), d(W[b + "norm1.bias"])),
                           qkv_w=d(pad_qkv(W[b + "attn.qkv.weight"])), qkv_b=d(pad_qkv(W[b + "attn.qkv.bias"])),
                           n2=(d(W[b + "norm2.weight"]), d(W[b + "norm2.bias"])),
                           fc1_w=d(W[b + "mlp.fc1.weight"]), fc1_b=d(W[b + "mlp.fc1.bias"]))
            blk.update(proj_w=d(pad_proj(W[b + "attn.proj.weight"])), proj_b=d(W[b + "attn.proj.bias"]), g1=d(W[b + "gamma_1"]),
                       fc2_w=d(W[b + "mlp.fc2.weight"]), fc2_b=d(W[b + "mlp.fc2.bias"]), g2=d(W[b + "gamma_2"]))
            self.vblocks.append(blk)
        # unit LayerNorm / RMSNorm parameters: a folded weight carries its norm's gain (and bias), so where no folded-norm
        # kernel takes a shape the norm runs as its own pass with gamma = 1, beta = 0 in front of the SAME weight
        self.unit_ln = (torch.ones(D, dtype=self.dtype, device=self.device), torch.zeros(D, dtype=self.dtype, device=self.device)) \
            if fold else None
        self.pj = dict(w1=d(W[PJ + "linear_1.weight"]), b1=d(W[PJ + "linear_1.bias"]),
                       w2=d(W[PJ + "linear_2.weight"]), b2=d(W[PJ + "linear_2.bias"]))
        sin, cos = _rope2d_tables(v)
        if hdp != hd:
            sin = torch.nn.functional.pad(sin, (0, hdp - hd), value=0.0).contiguous()
            cos = torch.nn.functional.pad(cos, (0, hdp - hd), value=1.0).contiguous()
        self.vit_sin, self.vit_cos = sin.to(self.device), cos.to(self.device)
        self.E = d(W[LM + "embed_tokens.weight"])
        self.lm_head = self.E if t.tie_word_embeddings or "mllm.lm_head.weight" not in W else d(W["mllm.lm_head.weight"])
        self.layers = []
        F = t.intermediate_size
        assert F % 16 == 0
        qkv_order = ops.llm_qkv_weight_order(t.head_dim, t.num_attention_heads, t.num_key_value_heads) \
            if (fold and self.LLM_QKV_EPILOGUE and t.head_dim in (64, 128)) else torch.arange((t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim)
        self.qkv_f_permuted = bool((qkv_order != torch.arange(qkv_order.numel())).any())
        for i in range(t.num_hidden_layers):
            b = f"{LM}layers.{i}."
            qkv = torch.cat([W[b + "self_attn.q_proj.weight"], W[b + "self_attn.k_proj.weight"],
                             W[b + "self_attn.v_proj.weight"]], dim=0)
            g, u = W[b + "mlp.gate_proj.weight"], W[b + "mlp.up_proj.weight"]
            # [gate16 | up16] row interleave expected by GAR_EPI_SWIGLU
            gu = torch.stack([g.view(F // 16, 16, -1), u.view(F // 16, 16, -1)], dim=1).reshape(2 * F, -1)
            ly = {}
            if fold:          # RMSNorm folded: W diag(g) — the tile GEMM's row_scale form, the decode GEMVs' norm_folded form
                # rows in the order the fused RoPE epilogue wants (identity for head_dim 64): self.qkv_f_permuted
                ly["qkv_f"] = d((qkv.float() * W[b + "input_layernorm.weight"].float()[None, :])[qkv_order])
                ly["gu_f"] = d(gu.float() * W[b + "post_attention_layernorm.weight"].float()[None, :])
            if plain:
                ly.update(ln1=d(W[b + "input_layernorm.weight"]), qkv=d(qkv), ln2=d(W[b + "post_attention_layernorm.weight"]),
                          gu=d(gu))
            ly.update(o=d(W[b + "self_attn.o_proj.weight"]), down=d(W[b + "mlp.down_proj.weight"]))
            self.layers.append(ly)
        self.unit_rms = torch.ones(t.hidden_size, dtype=self.dtype, device=self.device) if fold else None
        self.final_norm = d(W[LM + "norm.weight"])
        self.inv_freq = _llama_inv_freq(t)
        self.crop_ids_dev = torch.tensor(self.crop_tokens_ids, dtype=torch.int64, device=self.device)
        self._rope_cache = {}

    def _llm_rope(self, max_pos: int):
        if max_pos not in self._rope_cache:
            pos = torch.arange(max_pos, dtype=torch.float32)
            fr = pos[:, None] * self.inv_freq[None, :]
            self._rope_cache[max_pos] = (fr.cos().contiguous().to(self.device), fr.sin().contiguous().to(self.device))
        return self._rope_cache[max_pos]

    # Workspaces. Families whose shapes follow the request (tiles of a chunk, prompt length) are CAPACITY based: one flat
    # zero-filled allocation per (family, name) that only grows (x1.25) and is sliced into views, so an evaluation loop
    # with a different prompt length / tile count per item (gar_amd/bench_loops.py) does not allocate per shape.
    # Buffers a captured decode graph points at ("llm", "decode", "head") are keyed by their exact shape and never move;
    # the KV-cache states are kept in an LRU of MAX_LLM_STATES entries (their graphs go with them).
    _CAPACITY_FAMILIES = ("vit", "emb", "prefill")
    MAX_LLM_STATES = 2
    MAX_EOS_IDS = 16          # eos_token_id list entries the device-side stopping criterion holds (longer lists: host scan)
    SPLITK_MAX_ROWS = 64      # csrc/gemm.hip: split_k > 1 and gar_splitk_residual_rmsnorm are built for M <= 64 rows
    DOWN_SPLIT_K = 0          # K slices of the decode `down` GEMM at more than FUSE_NORM_MAX_BATCH rows: 0 = as many as give the
                              # two-weight-tile blocks of the skinny kernel a full grid — (hidden / 32) * slices >= 256: 4 at
                              # hidden 2048, 2 at 4096 (round 4: four row tiles re-read per ONE weight tile was 4 bytes of L2
                              # traffic per weight byte; GAR-1B 20.4 -> 18.7 us per layer with the reduce, GAR-8B 52.5 -> 44.1;
                              # unsplit 27.3 / 50.1; tools/bench_skinny.py); 1 = off

    def _buf(self, key: tuple, name: str, shape, dtype=None, zero=False):
        dtype = dtype or self.dtype
        if key[0] in self._CAPACITY_FAMILIES:
            n = 1
            for d in shape:
                n *= int(d)
            fam = self._ws.setdefault((key[0],), {})
            flat = fam.get(name)
            if flat is None or flat.dtype != dtype or flat.numel() < n:
                cap = n if flat is None or flat.dtype != dtype else max(n, int(flat.numel() * 1.25))
                flat = torch.zeros(cap, dtype=dtype, device=self.device)      # zero: padding rows must stay finite
                fam[name] = flat
            return flat[:n].view(*shape)
        d = self._ws.setdefault(key, {})
        t = d.get(name)
        if t is None or tuple(t.shape) != tuple(shape):
            t = (torch.zeros if zero else torch.empty)(shape, dtype=dtype, device=self.device)
            d[name] = t
        return t

    def workspace_bytes(self) -> int:
        """device bytes held by the cached workspaces (tests / soak checks)."""
        return sum(t.numel() * t.element_size() for d in self._ws.values() for t in d.values())

    # ---- vision tower + projector (A1-A6) -----------------------------------------------------------------------------
    @_on_model_device
    def get_image_features(self, pixel_values: torch.Tensor, global_mask_values: Optional[torch.Tensor] = None,
                           pooled: bool = True, out: Optional[torch.Tensor] = None,
                           mask_embeds: Optional[torch.Tensor] = None):
        """[Tt,3,H,W] (+ mask values of the same shape, still in the processor's [-1,1] encoding) -> [Tt, P*P, C_l].
        ``pooled=False`` stops after the projector and returns its [Tt * tokens, C_l] output (cls rows included),
        written into ``out`` when given. ``mask_embeds`` [Tt, C_v, g, g] (instead of ``global_mask_values``): the reference's
        form — the mask-embedding conv's output, added to the patch embeddings (modeling_perception_lm.py:195-196)."""
        cfg = self.config
        v = cfg.mllm_config.vision_config
        C_l = cfg.mllm_config.text_config.hidden_size
        if pixel_values.dim() == 5:
            pixel_values = pixel_values.flatten(0, 1)
        assert pixel_values.dim() == 4, f"pixel_values should be [tiles, 3, H, W], got {tuple(pixel_values.shape)}"
        pix = pixel_values.to(self.device, self.dtype).contiguous()
        msk = None
        if global_mask_values is not None:
            msk = global_mask_values.to(self.device, self.dtype).reshape(pix.shape).contiguous()
        Tt = pix.shape[0]
        n, D, Dm, H, hd = v.num_patches, v.embed_dim, v.mlp_dim, v.num_heads, self.v_hd     # hd: padded head dim
        Da = H * hd
        N = n + self.npt
        Npad = _round_up(N, 64)
        key = ("vit", Tt)
        x = self._buf(key, "x", (Tt, N, D))
        # zero-initialised once: the fused qkv GEMM writes the N real token rows only, rows N..Npad must stay finite
        Q = self._buf(key, "Q", (Tt, H, Npad, hd), zero=True)
        K = self._buf(key, "K", (Tt, H, Npad, hd), zero=True)
        att = self._buf(key, "att", (Tt * N, Da))
        qkv = Vt = vrow = None             # only the paths that need them allocate them (2.4 GB + 0.9 GB at 387 tiles)
        # bf16 at sizes the ping-pong GEMM takes: q / k leave the qkv GEMM already rotated, scaled and in attention
        # layout (GAR_EPI_QKV_ROPE), only V still needs its transpose; otherwise gemm + vit_qkv_post
        fused = self.dtype in HALF_DTYPES
        # bf16: V stays row-major [Tt, H, Npad, hd] (zero-initialised like Q / K: pad rows must be finite)
        Vr = self._buf(key, "Vr", (Tt, H, Npad, hd), zero=True) if fused and self.VIT_V_ROW_MAJOR else None
        f1 = self._buf(key, "f1", (Tt * N, max(Dm, C_l)))
        x2 = x.view(Tt * N, D)
        gathered = False
        if self.w_patch_gather is not None:
            # patches DMA'd from the image tiles into LDS by the GEMM itself: no im2col matrix (1216 columns per patch
            # written and re-read); the mask decode is one elementwise pass
            mb = self._buf(key, "maskbin", tuple(pix.shape))
            if msk is not None:
                ops.mask_decode(msk, mb, cfg.prompt_numbers)
            else:
                mb.zero_()
            gathered = ops.patch_embed(pix, mb, self.w_patch_gather, self.pos, x, v.patch_size, self.npt)
        if not gathered:
            A = self._buf(key, "im2col", (Tt * n, self.Kp))
            ops.patch_im2col(pix, msk, A, v.patch_size, cfg.prompt_numbers)
            ops.gemm(A, self.w_patch, x2, hip.EPI_PATCH_POS, pos=self.pos, tokens_in=n, tokens_out=N, token_offset=self.npt)
        if mask_embeds is not None:         # x = x + mask_embeds.flatten(2).transpose(1, 2) on the patch rows
            assert msk is None, "mask_embeds and global_mask_values are alternatives"
            me = mask_embeds.to(self.device, self.dtype)
            if tuple(me.shape) != (Tt, D, v.grid, v.grid):
                raise ValueError(f"mask_embeds {tuple(me.shape)} should be {(Tt, D, v.grid, v.grid)}")
            ops.tokens_add(x, me.flatten(2).transpose(1, 2).contiguous(), self.npt)
        if self.npt:
            ops.cls_pos_fill(x, self.cls, self.pos)
        ops.layernorm(x2, *self.norm_pre, v.ln_eps)
        q_scale = (v.head_dim ** -0.5) * LOG2E
        f1v = f1.view(-1)[:Tt * N * Dm].view(Tt * N, Dm)
        # Folded LayerNorms (bf16, passes large enough for the tile GEMM): qkv and fc1 read the residual stream x itself with
        # the folded weights and scale their accumulator rows by rstd; proj and fc2 write x's row statistics from their
        # epilogues; a one-thread-per-row kernel turns them into rstd. No LayerNorm pass, no normalised copy of x.
        M = Tt * N
        has_folded = "qkv_wf" in self.vblocks[0]
        use_folded_w = has_folded and (self.FOLD_NORMS or "qkv_w" not in self.vblocks[0])      # bf16 keeps only these by default
        fold = (self.FOLD_NORMS and fused and Vr is not None and has_folded and D % 64 == 0 and
                ops.tile_gemm_takes(M, 3 * Da, D, epilogue=hip.EPI_QKV_ROPE, row_scale=True) and
                ops.tile_gemm_takes(M, D, Da, epilogue=hip.EPI_BIAS_SCALE_RES, row_stats=True) and
                ops.tile_gemm_takes(M, Dm, D, epilogue=hip.EPI_BIAS_GELU, row_scale=True) and
                ops.tile_gemm_takes(M, D, Dm, epilogue=hip.EPI_BIAS_SCALE_RES, row_stats=True))
        # the cls key (row 0) enters through the softmax's initial state: 1 + 1024 keys are 16 kv tiles, not 17 (built for ONE
        # prefix row: a tower with cls + register tokens keeps them as ordinary keys)
        kv_prefix = 1 if (self.VIT_CLS_KEY_FOLD and self.npt == 1 and N > 1) else 0
        if fold:
            rstd = self._buf(key, "rstd", (M,), torch.float32)
            stats = self._buf(key, "stats", (M, D // 64, 2), torch.float32)
            ops.row_rstd(x2, v.ln_eps, False, rstd)                  # LN1 of block 0: its input came out of norm_pre
            for bi, blk in enumerate(self.vblocks):
                if not ops.gemm_qkv_rope(x2, blk["qkv_wf"], blk["qkv_bf"], att, Q, K, self.vit_sin, self.vit_cos, H, hd, N,
                                         Npad, self.npt, q_scale, V=Vr, row_scale=rstd):
                    raise hip.GarError("folded qkv GEMM refused a shape gar_gemm_tile_takes() accepted")
                ops.attention(Q, K, Vr, att, Tt, H, H, hd, N, Npad, N, Npad, causal=False, v_row_major=True, kv_prefix=kv_prefix)
                ops.gemm(att, blk["proj_w"], x2, hip.EPI_BIAS_SCALE_RES, bias=blk["proj_b"], residual=x2, gamma=blk["g1"],
                         row_stats=stats)
                ops.row_stats_finalize(stats, D, v.ln_eps, False, rstd)
                ops.gemm(x2, blk["fc1_wf"], f1v, hip.EPI_BIAS_GELU, bias=blk["fc1_bf"], row_scale=rstd)
                last = bi + 1 == len(self.vblocks)
                ops.gemm(f1v, blk["fc2_w"], x2, hip.EPI_BIAS_SCALE_RES, bias=blk["fc2_b"], residual=x2, gamma=blk["g2"],
                         row_stats=None if last else stats)
                if not last:
                    ops.row_stats_finalize(stats, D, v.ln_eps, False, rstd)
        hbuf = None if fold else self._buf(key, "h", (M, D))

        def ln_operand(blk, which):
            """stand-alone LayerNorm of the residual stream into hbuf + the (weight, bias) that goes with it: the plain pair
            behind the block's own gamma / beta, or — bf16 keeps only those — the FOLDED pair behind a unit LayerNorm
            ((x - mean) rstd: gamma, beta and the row centring live in the weight, LN(x) W^T + b = ((x - mean) rstd) Wc^T + b')."""
            if use_folded_w:
                ops.layernorm(x2, self.unit_ln[0], self.unit_ln[1], v.ln_eps, out=hbuf)
                return (blk["qkv_wf"], blk["qkv_bf"]) if which == 1 else (blk["fc1_wf"], blk["fc1_bf"])
            n = blk["n1" if which == 1 else "n2"]
            ops.layernorm(x2, n[0], n[1], v.ln_eps, out=hbuf)
            return (blk["qkv_w"], blk["qkv_b"]) if which == 1 else (blk["fc1_w"], blk["fc1_b"])

        for blk in ([] if fold else self.vblocks):
            qw, qb = ln_operand(blk, 1)
            if fused:
                if Vr is None and vrow is None:
                    qkv = self._buf(key, "qkv", (M, 3 * Da))
                    vrow = qkv.view(-1)[:M * Da].view(M, Da)
                    Vt = self._buf(key, "Vt", (Tt, H, hd, Npad))
                # with V= the row-major v output is not written (att stands in for the pointer the ABI wants)
                fused = ops.gemm_qkv_rope(hbuf, qw, qb, att if Vr is not None else vrow, Q, K,
                                          self.vit_sin, self.vit_cos, H, hd, N, Npad, self.npt, q_scale, V=Vr)
            if fused and Vr is None:
                ops.vit_v_transpose(vrow, Vt, Tt, N, H, hd, Npad)
            elif not fused:
                Vr = None
                if qkv is None:
                    qkv = self._buf(key, "qkv", (M, 3 * Da))
                if Vt is None:
                    Vt = self._buf(key, "Vt", (Tt, H, hd, Npad))
                ops.gemm(hbuf, qw, qkv, hip.EPI_BIAS, bias=qb)
                ops.vit_qkv_post(qkv, self.vit_sin, self.vit_cos, Q, K, Vt, Tt, N, self.npt, H, hd, Npad, q_scale)
            if Vr is not None:      # v left the qkv GEMM head-major like k: the attention transposes it on its LDS reads
                ops.attention(Q, K, Vr, att, Tt, H, H, hd, N, Npad, N, Npad, causal=False, v_row_major=True, kv_prefix=kv_prefix)
            else:
                ops.attention(Q, K, Vt, att, Tt, H, H, hd, N, Npad, N, Npad, causal=False)
            ops.gemm(att, blk["proj_w"], x2, hip.EPI_BIAS_SCALE_RES, bias=blk["proj_b"], residual=x2, gamma=blk["g1"])
            fw, fb = ln_operand(blk, 2)
            ops.gemm(hbuf, fw, f1v, hip.EPI_BIAS_GELU, bias=fb)
            ops.gemm(f1v, blk["fc2_w"], x2, hip.EPI_BIAS_SCALE_RES, bias=blk["fc2_b"], residual=x2, gamma=blk["g2"])
        # projector over all N tokens of a tile (cls row included, dropped by the pooling window)
        p1 = f1.view(-1)[:Tt * N * C_l].view(Tt * N, C_l)
        ops.gemm(x2, self.pj["w1"], p1, hip.EPI_BIAS_GELU, bias=self.pj["b1"])
        p2 = self._buf(key, "p2", (Tt * N, C_l)) if out is None or pooled else out
        assert tuple(p2.shape) == (Tt * N, C_l) and p2.is_contiguous()
        ops.gemm(p1, self.pj["w2"], p2, hip.EPI_BIAS, bias=self.pj["b2"])
        if cfg.mllm_config.projector_pooling_ratio != 2:
            raise hip.GarError("projector_pooling_ratio != 2 is not built")
        if not pooled:
            return p2                       # generate(): pooled on the way into the sequence (gar_pool_assemble)
        P = cfg.pooled_side
        feats = self._buf(key, "feats", (Tt, P * P, C_l))
        ops.pool2x2(p2, feats, v.grid, in_tile_tokens=N, in_token_offset=self.npt)
        return feats

    @_on_model_device
    def _mask_embed(self, binary_mask: torch.Tensor) -> torch.Tensor:
        """mask_patch_embedding(binary) (modeling_gar.py:326-328): [T,3,H,W] in {0,1} -> [T, C_v, g, g] (view of [T, g*g, C_v])."""
        v = self.config.mllm_config.vision_config
        D, n, g = v.embed_dim, v.num_patches, v.grid
        mb = binary_mask.to(self.device, self.dtype)
        if mb.dim() == 5:
            mb = mb.flatten(0, 1)
        mb = mb.contiguous()
        T = mb.shape[0]
        out = torch.empty(T, n, D, dtype=self.dtype, device=self.device)
        done = False
        if self.w_patch_gather is not None:
            if "gather" not in self._mask_only_w:       # the gather-ordered operand with the pixel tensor's slots zeroed + a zero pos table
                wg = self.w_patch_gather.clone().view(D, 2, -1)
                wg[:, 0] = 0
                self._mask_only_w["gather"] = (wg.view(D, -1), torch.zeros(n, D, dtype=self.dtype, device=self.device))
            wg, zpos = self._mask_only_w["gather"]
            done = ops.patch_embed(mb, mb, wg, zpos, out, v.patch_size, 0)       # pixel slots carry zero weights: any finite tensor does
        if not done:
            pp = v.patch_size * v.patch_size
            if "im2col" not in self._mask_only_w:       # the mask conv's weight in the pixel columns of the im2col operand
                w = torch.zeros_like(self.w_patch)
                w[:, :3 * pp] = self.w_patch[:, 3 * pp:6 * pp]
                self._mask_only_w["im2col"] = w
            A = torch.empty(T * n, self.Kp, dtype=self.dtype, device=self.device)
            ops.patch_im2col(mb, None, A, v.patch_size, self.config.prompt_numbers)
            ops.gemm(A, self._mask_only_w["im2col"], out.view(T * n, D))
        return out.view(T, g, g, D).permute(0, 3, 1, 2)

    # ---- inputs_embeds: embedding + placeholder scatter + RoI replay (A7-A11) -----------------------------------------
    @_on_model_device
    def build_inputs_embeds(self, input_ids, feats, bboxes, aspect_ratios, tiles_per_sample: int, validate=True,
                            video_frame_tokens: Optional[Sequence[int]] = None, proj: Optional[torch.Tensor] = None):
        """``feats`` [B*tiles, P*P, C] pooled features (embed_assemble + replay from the feature tensor), or — what
        ``generate`` uses — ``proj``: the un-pooled projector output of ``get_image_features(pooled=False)``; then the
        2x2 pool runs on the way into the sequence and the replay reads the pooled rows back from it (one read of the
        projector output + one write of the sequence per region: SURVEY.md section 8d's algorithmic bytes)."""
        cfg = self.config
        B, S = input_ids.shape
        C_l = cfg.mllm_config.text_config.hidden_size
        P = cfg.pooled_side
        key = ("emb", B, S)
        ids = input_ids.to(self.device, torch.int64).contiguous()
        if video_frame_tokens is not None:
            # A13 video replay (modeling_perception_lm.py:765-852): one pooled map per frame, crop token per frame
            crop_ids = [int(t) for t in video_frame_tokens]
            if len(crop_ids) != tiles_per_sample or len(crop_ids) > 8:
                raise hip.GarError(f"video replay: {len(crop_ids)} frame tokens for {tiles_per_sample} frames (max 8)")
            vkey = tuple(crop_ids)
            if vkey not in self._video_crop_ids:
                self._video_crop_ids[vkey] = torch.tensor(crop_ids, dtype=torch.int64, device=self.device)
            crop_ids_dev = self._video_crop_ids[vkey]
        else:
            crop_ids, crop_ids_dev = self.crop_tokens_ids, self.crop_ids_dev
        slot = self._buf(key, "slot", (B, S), torch.int32)
        counts = self._buf(key, "counts", (B,), torch.int32)
        spans = self._buf(key, "spans", (B, len(crop_ids), 2), torch.int32)
        embeds = self._buf(key, "embeds", (B, S, C_l))
        n_rows = tiles_per_sample * P * P
        rank_pos = None
        if proj is not None:
            rank_pos = self._buf(key, "rank_pos", (B, n_rows), torch.int32)
            rank_pos.zero_()                # ranks without a placeholder (count mismatch) must still index inside the row
        ops.placeholder_scan(ids, cfg.mllm_config.image_token_id, crop_ids_dev, slot, counts, spans, rank_pos)
        if proj is not None:
            v = cfg.mllm_config.vision_config
            ops.pool_assemble(ids, slot, self.E, proj, embeds, tiles_per_sample, v.grid, v.num_patches + self.npt, self.npt)
        else:
            ops.embed_assemble(ids, slot, self.E, feats, embeds, n_rows)
        if not validate:
            # the same conditions, evaluated on the device by ONE small kernel and OR-ed into a flag nobody has to wait for
            # (ADVICE r1 #5 / r2): what the kernels do with such inputs (clamped slots / ids, P*P rows written from the span
            # head) is defined but not what the reference computes. The per-sample "has a bbox for crop token c" bits are the
            # only host data (B int32, one small pinned upload).
            hb = self._upload([sum(1 << ci for ci, t in enumerate(crop_ids) if str(t) in bboxes[b]) for b in range(B)], torch.int32)
            if getattr(self, "_input_flags", None) is None:
                self._input_flags = torch.zeros(1, dtype=torch.int32, device=self.device)
            ops.input_check(ids, self.E.shape[0], counts, n_rows, spans, P * P, hb, self._input_flags)
        if validate:
            # reference errors (modeling_perception_lm.py:309-315, modeling_gar.py:356-360) need the counts on the host
            cnt = counts.tolist()
            sp = spans.tolist()
            for b in range(B):
                if cnt[b] != n_rows:
                    raise ValueError(f"Image features and image tokens do not match: tokens: {cnt[b]}, features {n_rows}")
        video = video_frame_tokens is not None
        if not video:
            ar = aspect_ratios.tolist() if torch.is_tensor(aspect_ratios) else aspect_ratios
        jobs = []
        for b in range(B):
            if video:
                ncw = nch = 1                       # each frame is its own P x P map (feat_h = feat_w = P, :787)
            else:
                ncw, nch = int(ar[b][0]), int(ar[b][1])
                assert ncw * nch == tiles_per_sample - 1, f"{ncw * nch} != {tiles_per_sample - 1}"
            feat_h, feat_w = P * nch, P * ncw
            for ci, crop_token in enumerate(crop_ids):
                if str(crop_token) not in bboxes[b]:
                    if validate and sp[b][ci][1] >= 0:
                        raise KeyError(str(crop_token))
                    continue
                if validate:
                    if sp[b][ci][1] < 0:
                        continue            # token not in input_ids: the reference skips it (:356)
                    if sp[b][ci][1] - sp[b][ci][0] + 1 != P * P:
                        raise ValueError(f"crop token {crop_token} spans {sp[b][ci]} but the replay is {P * P} rows")
                # box math of modeling_gar.py:366-387 in Python floats, then fp32 like torch.tensor(..., float32)
                x1, y1, x2, y2 = [float(z) for z in bboxes[b][str(crop_token)]]
                orig_h, orig_w = feat_h * cfg.feat_stride, feat_w * cfg.feat_stride
                ss = feat_w / orig_w
                roi = (x1 * orig_w * ss, y1 * orig_h * ss, x2 * orig_w * ss, y2 * orig_h * ss)
                # image: map = tiles 1.. (thumbnail dropped, modeling_gar.py:351); video: map = frame ci only
                jobs.append((b, ci, ci if video else 1, ncw, nch, *roi, ss))
        if jobs:         # every crop token of every sample in one launch (the roi table is the only H2D copy)
            jt = ops.roi_jobs_tensor(jobs, self.device)
            if proj is not None:
                ops.roi_replay_inplace(embeds, spans, rank_pos, jt, len(crop_ids), P, C_l, S, 2, True)
            else:
                ops.roi_replay_batched(feats, embeds, spans, jt, len(crop_ids), tiles_per_sample, P, C_l, S, 2, True)
        return embeds

    # ---- Llama (A12) --------------------------------------------------------------------------------------------------
    def _llm_state(self, B: int, Smax: int, slot: int = 0):
        """KV cache + counters of one (B, Smax) bucket. ``slot``: independent copies of the same bucket — the pipelined driver
        (generate_pipelined) decodes batch i from one slot while batch i + 1 prefills into the other."""
        t = self.config.mllm_config.text_config
        key = ("llm", B, Smax, slot)
        if key in self._llm_lru:
            self._llm_lru.remove(key)
        self._llm_lru.append(key)
        while sum(1 for k in self._llm_lru if k[3] == slot) > self.MAX_LLM_STATES:        # MAX_LLM_STATES buckets per slot
            old = next(k for k in self._llm_lru if k[3] == slot)
            self._llm_lru.remove(old)
            self._ws.pop(old, None)
            for gk in [g for g in self._graphs if g[0] == old]:
                del self._graphs[gk]
            # the per-batch decode / head workspaces (logits [B, vocab], attention partials, split-K slices) go with the
            # last live state of that batch size: a service that sees many distinct B does not accumulate them
            if not any(k[1] == old[1] for k in self._llm_lru):
                for wk in [w for w in self._ws if w[0] in ("decode", "head") and w[1] == old[1]]:
                    self._ws.pop(wk, None)
        L, Hkv, hd = t.num_hidden_layers, t.num_key_value_heads, t.head_dim
        st = dict(
            Kc=self._buf(key, "Kc", (L, B, Hkv, Smax, hd), zero=True),
            Vc=self._buf(key, "Vc", (L, B, Hkv, Smax, hd), zero=True),      # row-major like K: an append is one row
            counters=self._buf(key, "counters", (4,), torch.int32, zero=True),   # [pos, kv_len, step, -]
            cur=self._buf(key, "cur", (B,), torch.int64, zero=True),
            # first real row of every sequence (left-padded batch; zeros otherwise): read by the qkv-post and attention
            # kernels of the prefill and of the captured decode step, so one graph serves padded and unpadded requests
            left_pad=self._buf(key, "left_pad", (B,), torch.int32, zero=True),
            # the greedy loop's stopping criterion, evaluated by the argmax kernel (gar_argmax): the eos ids of the request
            # (-1 = unused entry), the step at which a row first produced one (-1 = running), the number of finished rows
            eos_ids=self._buf(key, "eos_ids", (self.MAX_EOS_IDS,), torch.int64),
            finished=self._buf(key, "finished", (B,), torch.int32),
            done_count=self._buf(key, "done_count", (1,), torch.int32, zero=True),
            # do_sample = True (gar_sample): [temperature, top_p, top_k] and the Philox key of the request, read by the captured step
            sample_params=self._buf(key, "sample_params", (4,), torch.float32, zero=True),
            sample_seed=self._buf(key, "sample_seed", (1,), torch.int64, zero=True),
            sampling=False,
            slot=slot,
        )
        return key, st

    def _prefill(self, embeds: torch.Tensor, st, Smax: int, b0: int = 0):
        """Prefill of one chunk of sequences [B,S,C] whose KV goes to rows b0..b0+B of the shared cache. Returns the last
        prompt row of every sequence after the last layer ([B, C] row-strided view): what the head reads."""
        t = self.config.mllm_config.text_config
        B, S, C_l = embeds.shape
        Hq, Hkv, hd, F = t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.intermediate_size
        key = ("prefill", B, S)
        h = embeds.view(B * S, C_l)
        M = B * S
        qd = (Hq + 2 * Hkv) * hd
        Spad = _round_up(S, 64)
        Q = self._buf(key, "Q", (B, Hq, Spad, hd))
        cos, sin = self._llm_rope(Smax)
        q_scale = (hd ** -0.5) * LOG2E
        lp = st["left_pad"][b0:b0 + B]
        bf16 = self.dtype in HALF_DTYPES
        has_folded = "qkv_f" in self.layers[0]
        use_folded_w = has_folded and (self.FOLD_NORMS or "qkv" not in self.layers[0])         # bf16 keeps only these by default
        # Folded RMSNorms (bf16, tile-GEMM sized passes): qkv and gate/up read the residual stream h with W diag(g) and scale
        # their accumulator rows by rstd; o and down write h's row sums of squares from their epilogues (see get_image_features)
        fold = (self.FOLD_NORMS and bf16 and has_folded and C_l % 64 == 0 and
                ops.tile_gemm_takes(M, qd, C_l, row_scale=True) and
                ops.tile_gemm_takes(M, C_l, Hq * hd, epilogue=hip.EPI_RES, row_stats=True) and
                ops.tile_gemm_takes(M, 2 * F, C_l, epilogue=hip.EPI_SWIGLU, row_scale=True) and
                ops.tile_gemm_takes(M, C_l, F, epilogue=hip.EPI_RES, row_stats=True))
        rstd = stats = xn = None
        if fold:
            rstd = self._buf(key, "rstd", (M,), torch.float32)
            stats = self._buf(key, "stats", (M, C_l // 64, 2), torch.float32)
            ops.row_rstd(h, t.rms_norm_eps, True, rstd)                  # input_layernorm of layer 0: h came from the embedding pass
        else:
            xn = self._buf(key, "xn", (M, C_l))                          # the normalised copy of h: only without folded norms
        fused_qkv = fold and self.LLM_QKV_EPILOGUE and hd in (64, 128)
        strip = use_folded_w and self.qkv_f_permuted        # q / k head columns of a qkv_f product are in the fused epilogue's order
        L = len(self.layers)
        prune = self.PRUNE_LAST_PREFILL_LAYER and S > 1
        att = ff = None
        if L > 1 or not prune:
            att = self._buf(key, "att", (M, Hq * hd))
            ff = self._buf(key, "ff", (M, F))
        for li, ly in enumerate(self.layers):
            last = li + 1 == L
            Kc, Vc = st["Kc"][li][b0:b0 + B], st["Vc"][li][b0:b0 + B]         # this chunk's rows of the shared cache
            # ---- q / k / v of EVERY row (the decode steps read this layer's K / V rows too)
            if fused_qkv:
                if not ops.gemm_qkv_rope_llm(h, ly["qkv_f"], Q, Kc, Vc, cos, sin, B, S, Spad, Hq, Hkv, hd, Smax, 0, None,
                                             q_scale, left_pad=lp, row_scale=rstd):
                    raise hip.GarError("fused qkv GEMM refused a shape gar_gemm_tile_takes() accepted")
            else:
                qb = self._qkv_buf(key, B, S, Hq, Hkv, hd)
                if fold:
                    ops.gemm(h, ly["qkv_f"], qb, row_scale=rstd)
                elif use_folded_w:      # unit-gain RMSNorm pass + the folded weight (its gain is in W)
                    ops.rmsnorm(h, self.unit_rms, t.rms_norm_eps, out=xn)
                    ops.gemm(xn, ly["qkv_f"], qb)
                else:
                    ops.rmsnorm(h, ly["ln1"], t.rms_norm_eps, out=xn)
                    ops.gemm(xn, ly["qkv"], qb)
                ops.llm_qkv_post(qb, cos, sin, Q, Kc, Vc, B, S, Spad, Hq, Hkv, hd, Smax, 0, None, q_scale, left_pad=lp,
                                 strip_order=strip)
            if last and prune:
                return self._prefill_tail(ly, h, Q, Kc, Vc, st, key, B, S, Spad, Smax, lp)
            ops.attention(Q, Kc, Vc, att, B, Hq, Hkv, hd, S, Spad, S, Smax, causal=True, kv_start=lp, v_row_major=True)
            if fold:
                ops.gemm(att, ly["o"], h, hip.EPI_RES, residual=h, row_stats=stats)
                ops.row_stats_finalize(stats, C_l, t.rms_norm_eps, True, rstd)
                ops.gemm(h, ly["gu_f"], ff, hip.EPI_SWIGLU, row_scale=rstd)
                ops.gemm(ff, ly["down"], h, hip.EPI_RES, residual=h, row_stats=None if last else stats)
                if not last:
                    ops.row_stats_finalize(stats, C_l, t.rms_norm_eps, True, rstd)
            else:
                ops.gemm(att, ly["o"], h, hip.EPI_RES, residual=h)
                if use_folded_w:
                    ops.rmsnorm(h, self.unit_rms, t.rms_norm_eps, out=xn)
                    ops.gemm(xn, ly["gu_f"], ff, hip.EPI_SWIGLU)
                else:
                    ops.rmsnorm(h, ly["ln2"], t.rms_norm_eps, out=xn)
                    ops.gemm(xn, ly["gu"], ff, hip.EPI_SWIGLU)
                ops.gemm(ff, ly["down"], h, hip.EPI_RES, residual=h)
        return h.view(B, S, C_l)[:, S - 1, :]                                   # row-strided view [B, C]

    def _prefill_tail(self, ly, h, Q, Kc, Vc, st, key, B, S, Spad, Smax, lp):
        """The last layer of a prefill behind its qkv GEMM, for the LAST prompt row of each of the B sequences only
        (PRUNE_LAST_PREFILL_LAYER): single-row attention over the sequence's S cache rows with the split-KV decode kernel —
        the query is read in place from row S - 1 of Q [B, Hq, Spad, hd] — then o / RMSNorm / gate-up / down as B-row GEMMs on
        the rows h[b, S - 1] (row-strided views, updated in place like the full-width layers update h)."""
        t = self.config.mllm_config.text_config
        C_l = h.shape[1]
        Hq, Hkv, hd, F = t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.intermediate_size
        hl = h.view(B, S, C_l)[:, S - 1, :]
        attl = self._buf(key, "att_last", (B, Hq * hd))
        ffl = self._buf(key, "ff_last", (B, F))
        kv_len = st["counters"][3:4]                    # scratch word of the counters: the prompt length, for the kernel's kv_len_dev
        kv_len.fill_(S)
        nsplit = max(1, min(self.DECODE_ATTN_MAX_SPLITS, self.DECODE_ATTN_BLOCKS // max(1, B * Hkv)))
        dws = self._buf(key, "attn_ws_last", (ops.attention_decode_workspace(B, Hq, hd, nsplit),), torch.uint8)
        ops.attention_decode(Q[:, :, S - 1], Kc, Vc, attl, B, Hq, Hkv, hd, Smax, kv_len, nsplit, dws, kv_start=lp,
                             q_stride=Spad * hd)
        ops.gemm(attl, ly["o"], hl, hip.EPI_RES, residual=hl)
        if "gu_f" in ly and (self.FOLD_NORMS or "gu" not in ly):
            if B <= 64:     # RMSNorm inside the GEMM: gain in the weight, row sums of squares off the matrix pipe
                ops.gemm(hl, ly["gu_f"], ffl, hip.EPI_SWIGLU, norm_folded=True, norm_eps=t.rms_norm_eps)
            else:
                xnl = self._buf(key, "xn_last", (B, C_l))
                ops.rmsnorm(hl, self.unit_rms, t.rms_norm_eps, out=xnl)
                ops.gemm(xnl, ly["gu_f"], ffl, hip.EPI_SWIGLU)
        else:
            xnl = self._buf(key, "xn_last", (B, C_l))
            ops.rmsnorm(hl, ly["ln2"], t.rms_norm_eps, out=xnl)
            ops.gemm(xnl, ly["gu"], ffl, hip.EPI_SWIGLU)
        ops.gemm(ffl, ly["down"], hl, hip.EPI_RES, residual=hl)
        return hl

    def _qkv_buf(self, key, B, S, Hq, Hkv, hd):
        """the [B*S, (Hq + 2 Hkv) hd] qkv GEMM output of the unfused prefill path (f32 / FOLD_NORMS off): lazily allocated"""
        return self._buf(key, "qkv", (B * S, (Hq + 2 * Hkv) * hd))

    def _upload(self, values, dtype, cache: bool = True) -> torch.Tensor:
        """small host constants of a request (eos ids, has-bbox bits, sampling parameters) as a device tensor. The values repeat from
        call to call (one eos set, one crop-token pattern per evaluation loop): their device copies are cached by value — no pinned
        host allocation and no upload per generate() (visible at batch 1). A cache miss is one pageable H2D copy."""
        key = (dtype, tuple(values))
        memo = self.__dict__.setdefault("_upload_cache", {})
        t = memo.get(key) if cache else None
        if t is None:
            t = torch.tensor(list(values), dtype=dtype).to(self.device)
            if cache:
                if len(memo) > 256:
                    memo.clear()
                memo[key] = t
        return t

    def _head(self, last_rows: torch.Tensor, B: int, out_tokens, st, cur=None, normed: Optional[torch.Tensor] = None,
              finished=None, row0: int = 0):
        """final RMSNorm + lm_head + greedy argmax of the given [B, C] rows (row-strided view allowed). ``normed``: the
        rows after the final norm, when the caller's last launch produced them already (split-K decode path)."""
        cur = st["cur"] if cur is None else cur
        t = self.config.mllm_config.text_config
        C_l, V = t.hidden_size, t.vocab_size
        key = ("head", B, st.get("slot", 0))       # per slot: a pipelined prefill's first-token head runs beside the other slot's decode
        xn = self._buf(key, "xn", (B, C_l))
        Vld = _round_up(V, 64)
        logits = self._buf(key, "logits", (B, Vld))
        ws = self._buf(key, "amws", (ops.argmax_workspace(B, V),), torch.uint8)
        if normed is not None:
            ops.gemm(normed, self.lm_head, logits)
        elif B <= 16:       # RMSNorm folded into the GEMV prologue; for more rows one tiny norm launch is cheaper
            ops.gemm(last_rows, self.lm_head, logits, norm_w=self.final_norm, norm_eps=t.rms_norm_eps)
        else:
            ops.rmsnorm(last_rows, self.final_norm, t.rms_norm_eps, out=xn)
            ops.gemm(xn, self.lm_head, logits)
        fin = st["finished"] if finished is None else finished
        if st.get("sampling"):
            ops.sample(logits, V, out_tokens, out_tokens.stride(0), st["counters"][2:3], cur, st["sample_params"], st["sample_seed"],
                       eos_ids=st["eos_ids"], finished=fin, done_count=st["done_count"], row_offset=row0)
        else:
            ops.argmax(logits, V, out_tokens, out_tokens.stride(0), st["counters"][2:3], cur, ws, eos_ids=st["eos_ids"],
                       finished=fin, done_count=st["done_count"])
        return logits

    def _decode_step(self, st, B: int, Smax: int, out_tokens):
        t = self.config.mllm_config.text_config
        C_l = t.hidden_size
        Hq, Hkv, hd, F = t.num_attention_heads, t.num_key_value_heads, t.head_dim, t.intermediate_size
        key = ("decode", B)
        h = self._buf(key, "h", (B, C_l))
        xn = self._buf(key, "xn", (B, C_l))
        qkv = self._buf(key, "qkv", (B, (Hq + 2 * Hkv) * hd))
        Q = self._buf(key, "Q", (B, Hq, 1, hd))
        att = self._buf(key, "att", (B, Hq * hd))
        ff = self._buf(key, "ff", (B, F))
        cos, sin = self._llm_rope(Smax)
        q_scale = (hd ** -0.5) * LOG2E
        pos_dev, kvlen_dev = st["counters"][0:1], st["counters"][1:2]
        nsplit = max(1, min(self.DECODE_ATTN_MAX_SPLITS, self.DECODE_ATTN_BLOCKS // max(1, B * Hkv)))
        dws = self._buf(key, "attn_ws", (ops.attention_decode_workspace(B, Hq, hd, nsplit),), torch.uint8)
        ops.embed_lookup(st["cur"], self.E, h)
        bf16 = self.dtype in HALF_DTYPES
        use_folded_w = "qkv_f" in self.layers[0] and (self.FOLD_NORMS or "qkv" not in self.layers[0])
        # how the two RMSNorms of a layer reach their GEMVs:
        #   folded  (bf16, B <= 64): gain in the weight (qkv_f / gu_f), row sums of squares off the matrix pipe inside the GEMV
        #           (gar_gemm_params.norm_folded) — no norm launch, no normalised copy, ONE copy of the weights
        #   fuse    (plain weights, B <= 16): x * g in the GEMV prologue (gar_gemm_params.norm_w)
        #   else    a norm launch in front of the GEMV (unit gain when only the folded weights exist)
        folded = use_folded_w and bf16 and B <= 64
        fuse = not use_folded_w and B <= self.FUSE_NORM_MAX_BATCH
        strip = use_folded_w and self.qkv_f_permuted
        # `down` (K = intermediate size, only hidden/16 weight tiles) streams from DOWN_SPLIT_K x the workgroups as K slices
        # whose fp32 products are reduced — with the residual add and, on the last layer, the final RMSNorm — by the launch that
        # follows anyway (gar_gemm's split_k and gar_splitk_residual_rmsnorm take at most SPLITK_MAX_ROWS rows; larger batches
        # keep EPI_RES)
        want = self.DOWN_SPLIT_K or (4 if C_l <= 2048 else 2)
        split = want if (B > self.FUSE_NORM_MAX_BATCH and B <= self.SPLITK_MAX_ROWS and bf16
                         and F % (64 * want) == 0 and C_l <= 4096) else 1
        partial = self._buf(key, "down_partial", (split, B, C_l), torch.float32) if split > 1 else None
        normed = None
        gu_folded = folded and (self.DECODE_GU_NORM_FOLDED or "gu" not in self.layers[0])
        qkv_folded = folded and (self.DECODE_GU_NORM_FOLDED or "qkv" not in self.layers[0])
        L = len(self.layers)
        xn_ready = False            # xn holds RMSNorm(h; this layer's input_layernorm) — written by the previous layer's reduce
        for li, ly in enumerate(self.layers):
            if qkv_folded:
                ops.gemm(h, ly["qkv_f"], qkv, norm_folded=True, norm_eps=t.rms_norm_eps)
            elif fuse:
                ops.gemm(h, ly["qkv"], qkv, norm_w=ly["ln1"], norm_eps=t.rms_norm_eps)
            else:
                if not xn_ready:
                    ops.rmsnorm(h, self.unit_rms if use_folded_w else ly["ln1"], t.rms_norm_eps, out=xn)
                ops.gemm(xn, ly["qkv_f"] if use_folded_w else ly["qkv"], qkv)
            # bf16: RoPE, q scale and the cache append run inside the attention launch (one launch instead of two)
            if not (self.DECODE_ATTN_TAKES_QKV and
                    ops.attention_decode_qkv(qkv, cos, sin, st["Kc"][li], st["Vc"][li], att, B, Hq, Hkv, hd, Smax, pos_dev,
                                             q_scale, nsplit, dws, left_pad=st["left_pad"], strip_order=strip)):
                ops.llm_qkv_post(qkv, cos, sin, Q, st["Kc"][li], st["Vc"][li], B, 1, 1, Hq, Hkv, hd, Smax, 0, pos_dev, q_scale,
                                 left_pad=st["left_pad"], strip_order=strip)
                ops.attention_decode(Q, st["Kc"][li], st["Vc"][li], att, B, Hq, Hkv, hd, Smax, kvlen_dev, nsplit, dws,
                                     kv_start=st["left_pad"])
            ops.gemm(att, ly["o"], h, hip.EPI_RES, residual=h)
            if gu_folded:       # W diag(g) + row sums of squares off the matrix pipe: no RMSNorm launch, no normalised copy
                ops.gemm(h, ly["gu_f"], ff, hip.EPI_SWIGLU, norm_folded=True, norm_eps=t.rms_norm_eps)
            elif fuse:
                ops.gemm(h, ly["gu"], ff, hip.EPI_SWIGLU, norm_w=ly["ln2"], norm_eps=t.rms_norm_eps)
            else:
                ops.rmsnorm(h, self.unit_rms if use_folded_w else ly["ln2"], t.rms_norm_eps, out=xn)
                ops.gemm(xn, ly["gu_f"] if use_folded_w else ly["gu"], ff, hip.EPI_SWIGLU)
            xn_ready = False
            if split > 1:
                ops.gemm(ff, ly["down"], None, partial=partial)
                lastl = li + 1 == L
                # h += down (one rounding, like EPI_RES); the reduce also writes the RMSNorm the next GEMV wants when that GEMV
                # does not take it folded: the final norm for the head, or the next layer's input_layernorm
                if lastl:
                    ops.splitk_residual_rmsnorm(partial, h, self.final_norm, t.rms_norm_eps, out=xn)
                    normed = xn
                elif qkv_folded:
                    ops.splitk_residual_rmsnorm(partial, h, None, t.rms_norm_eps, out=None)
                else:
                    nxt = self.unit_rms if use_folded_w else self.layers[li + 1]["ln1"]
                    ops.splitk_residual_rmsnorm(partial, h, nxt, t.rms_norm_eps, out=xn)
                    xn_ready = True
            else:
                ops.gemm(ff, ly["down"], h, hip.EPI_RES, residual=h)
        logits = self._head(h, B, out_tokens, st, normed=normed)
        ops.counter_add(st["counters"][0:3], 1)
        return logits

    # ---- pass planning --------------------------------------------------------------------------------------------
    VIT_CHUNK_ROWS = 400_000      # row caps of one vision-tower / prefill pass (activation workspaces: ~28 KB resp.
    PREFILL_CHUNK_ROWS = 131_072  # ~38 KB per row for GAR-1B); the 4-GiB operand limit of the tile GEMM caps them too

    def _plan_passes(self, B: int, tiles: int, S: int):
        """(image tiles per vision-tower pass, sequences per prefill pass), see gar_amd/planner.py."""
        if self.prefill_chunk is not None:
            c = max(1, min(B, self.prefill_chunk))
            seq = [min(c, B - b) for b in range(0, B, c)]
            return [n * tiles for n in seq], seq
        cfg = self.config
        v, t = cfg.mllm_config.vision_config, cfg.mllm_config.text_config
        cus = hip.num_cus(self.device.index or 0)
        Da = v.num_heads * self.v_hd
        vit_gemms = [(3 * Da, v.embed_dim), (v.embed_dim, Da), (v.mlp_dim, v.embed_dim), (v.embed_dim, v.mlp_dim)]
        lim = (1 << 31) - 4096                 # A operand bytes / 2 addressable by the GEMM's buffer descriptor
        vrows = min(self.VIT_CHUNK_ROWS, lim // max(v.mlp_dim, self.Kp, 3 * Da))
        tile_chunks = plan_chunks(B * tiles, v.num_patches + self.npt, vit_gemms, vrows, cus) if tiles else []
        qd = (t.num_attention_heads + 2 * t.num_key_value_heads) * t.head_dim
        llm_gemms = [(qd, t.hidden_size), (t.hidden_size, t.num_attention_heads * t.head_dim),
                     (2 * t.intermediate_size, t.hidden_size), (t.hidden_size, t.intermediate_size)]
        lrows = min(self.PREFILL_CHUNK_ROWS, lim // max(t.intermediate_size, qd))
        seq_chunks = plan_chunks(B, S, llm_gemms, lrows, cus)
        return tile_chunks, seq_chunks

    # ---- generate -------------------------------------------------------------------------------------------------
    def generate(self, *args, **kwargs) -> GenerateOutput:
        """Greedy region captioning, reference semantics of GARModel.generate (modeling_gar.py:295-428): the prompt phase
        (:meth:`generate_begin`: vision tower, sequence assembly + RoI replay, prefill, first token) followed by the decode
        loop (:meth:`generate_finish`) on the current stream. See generate_begin for the arguments."""
        return self.generate_finish(self.generate_begin(*args, **kwargs))

    def generate_pipelined(self, samples, **kwargs) -> List[GenerateOutput]:
        """``[generate(**s, **kwargs) for s in samples]`` through a :class:`GenerationPipeline`: the decode loop of batch i
        runs on a second HIP stream beside the prompt phase of batch i + 1. Bit-identical to the sequential calls."""
        pipe = GenerationPipeline(self, **kwargs)
        outs = []
        for smp in samples:
            outs.extend(pipe.submit(smp))
        outs.extend(pipe.flush())
        return outs

    @_on_model_device
    @torch.no_grad()
    def generate_begin(self, pixel_values=None, global_mask_values=None, aspect_ratios=None, bboxes=None, input_ids=None,
                       attention_mask=None, generation_config=None, output_hidden_states=None, return_dict=None,
                       max_new_tokens: Optional[int] = None, eos_token_id=None, use_graph: bool = True, validate: bool = True,
                       return_logits: bool = False, sync_every: int = 16, feature_replay_video: bool = False,
                       video_frame_tokens: Optional[Sequence[int]] = None, forced_tokens: Optional[torch.Tensor] = None,
                       state_slot: int = 0, seed: Optional[int] = None, **generate_kwargs) -> "_PendingGeneration":
        """Region captioning, reference semantics of GARModel.generate (modeling_gar.py:295-428): greedy search, or — with
        ``do_sample=True`` in the generation config — temperature / top-k / top-p sampling on the device (``seed``: the Philox key of
        the request; None draws one from torch's global generator).

        B = input_ids.shape[0] samples are processed together (the reference handles B=1 per call; its loop over
        ``batch_idx`` is kept). ``pixel_values`` / ``global_mask_values``: [B*(T+1), 3, H, W] (flattened tiles as the
        reference's callers pass them) or [B, T+1, 3, H, W].

        ``forced_tokens`` [B, n] (tests): teacher forcing — ``sequences`` / ``logits`` still report what the model chose at
        every step, but the token fed back as step j's input is ``forced_tokens[:, j]``, so two models (bf16 vs f32) can be
        compared on identical contexts over a whole caption."""
        gc = generation_config
        sampling = None
        if generate_kwargs:
            # the reference forwards **generate_kwargs to HF's generate (modeling_gar.py:418-426), which lays them OVER the generation
            # config: the same here — an option passed as a keyword is read (and refused if not built) like one in the config, never
            # swallowed; a keyword that is no generation option at all is an error as it is in HF
            unknown = sorted(k for k in generate_kwargs if k not in _GENERATION_KWARGS)
            if unknown:
                raise TypeError(f"generate() got unexpected keyword arguments {unknown} (generation options: see GenerationConfig)")
            base = gc
            base_get = (lambda k, d=None: d) if base is None else \
                ((lambda k, d=None: base.get(k, d)) if isinstance(base, dict) else (lambda k, d=None: getattr(base, k, d)))
            gc = _OverlaidOptions(base_get, dict(generate_kwargs))
        if gc is not None:
            get = gc.get if isinstance(gc, (dict, _OverlaidOptions)) else (lambda k, d=None: getattr(gc, k, d))
            _refuse_non_greedy(get)
            if get("do_sample", False):
                sampling = _sampling_options(get)
            max_new_tokens = max_new_tokens or get("max_new_tokens")
            if eos_token_id is None:
                eos_token_id = get("eos_token_id")
            pad_token_id = get("pad_token_id")
        else:
            pad_token_id = None
        max_new_tokens = int(max_new_tokens or 64)
        cfg = self.config
        B, S = input_ids.shape
        # attention_mask is forwarded to the Llama generate by the reference (modeling_gar.py:418-426). A LEFT-padded batch
        # (HF's convention for generation: prompts of different lengths right-aligned, zeros in front) is supported: a
        # sequence keeps its rows in the padded layout, its RoPE positions count from its first real token and the padding
        # keys stay hidden (HF: position_ids = cumsum(mask) - 1, causal mask AND attention_mask).
        left_pad = am_dev = None
        if attention_mask is not None:
            am = (attention_mask != 0)
            if tuple(am.shape) != (B, S):
                raise ValueError(f"attention_mask {tuple(am.shape)} does not match input_ids {(B, S)}")
            if validate:
                amc = am.cpu()
                if not bool(amc.all()):
                    if not bool((amc[:, 1:] >= amc[:, :-1]).all()) or not bool(amc[:, -1].all()):
                        raise hip.GarError("attention_mask must be LEFT-padded (zeros only in front of every prompt): a "
                                           "right-padded row would be continued after its padding")
                    left_pad = (S - amc.sum(1)).to(torch.int32).to(self.device)
            else:                      # no host sync: the zero count is the pad; the device-side input check (gar_input_check)
                am_dev = am.to(self.device).contiguous()       # raises INPUT_MASK_NOT_LEFT_PADDED for a row that is not 0...01...1
                left_pad = (S - am_dev.sum(1)).to(torch.int32)
        # KV capacity in buckets of 256 positions: evaluation loops with a different prompt length per item reuse one
        # cache, one token buffer ([B, Smax], sliced) and one captured decode graph per bucket
        if forced_tokens is not None:
            forced_tokens = forced_tokens.to(self.device, torch.int64)
        Smax = _round_up(S + max_new_tokens, 256)
        skey, st = self._llm_state(B, Smax, state_slot)
        out_tokens = self._buf(skey, "out_tokens", (B, Smax), torch.int64, zero=True)[:, :max_new_tokens]
        st["counters"].zero_()
        if left_pad is None:
            st["left_pad"].zero_()
        else:
            st["left_pad"].copy_(left_pad)
        eos_list = []
        if eos_token_id is not None:
            eos_list = [int(e) for e in (eos_token_id if isinstance(eos_token_id, (list, tuple)) else [eos_token_id])]
        # the stopping criterion lives on the device (gar_argmax latches each row's first eos step): fills take their value as a
        # kernel argument, the id table is one small upload
        eos_on_device = len(eos_list) <= self.MAX_EOS_IDS
        st["finished"].fill_(-1)
        st["done_count"].zero_()
        st["eos_ids"].fill_(-1)
        if eos_list and eos_on_device:
            st["eos_ids"][:len(eos_list)].copy_(self._upload(eos_list, torch.int64))
        st["sampling"] = sampling is not None
        if sampling is not None:
            if seed is None:          # HF draws from torch's global generator: so does the key of this request (torch.manual_seed reproduces it)
                seed = int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64))
                if torch.distributed.is_available() and torch.distributed.is_initialized():
                    # identically seeded data-parallel ranks must not replay one stream on different regions
                    seed = (seed ^ (0x9E3779B97F4A7C15 * (torch.distributed.get_rank() + 1))) & (2 ** 62 - 1)
            st["sample_params"].copy_(self._upload([sampling[0], sampling[1], float(sampling[2]), 0.0], torch.float32))
            st["sample_seed"].copy_(self._upload([int(seed)], torch.int64, cache=False))
        V = cfg.mllm_config.text_config.vocab_size
        tiles = 0
        if pixel_values is not None:
            tiles = pixel_values.shape[0] // B if pixel_values.dim() == 4 else pixel_values.shape[1]
            pv = pixel_values.reshape(B, tiles, *pixel_values.shape[-3:])
            gm = None if global_mask_values is None else global_mask_values.reshape(pv.shape)
            if feature_replay_video and video_frame_tokens is None:
                # <|reserved_special_token_{2+f}|> of frame f (modeling_perception_lm.py:777-780): the first
                # prompt_numbers ids are config.crop_tokens_ids, the following reserved tokens are consecutive ids
                base = list(self.crop_tokens_ids)
                video_frame_tokens = (base + [base[-1] + 1 + i for i in range(max(0, tiles - len(base)))])[:tiles]
        # The vision tower runs over chunks of image tiles and the prefill over chunks of sequences (bounded activation
        # memory, GEMM operands < 4 GiB; sizes from _plan_passes); the projector output of the whole batch is kept, and
        # the decode loop below then serves all B sequences of the shared KV cache in one weight pass per token.
        first_logits = []
        self._input_flags = torch.zeros(1, dtype=torch.int32, device=self.device)
        if am_dev is not None:      # validate=False: the mask's left-padded form is checked on the device, flag read by the caller
            ops.input_check(input_ids.to(self.device, torch.int64).contiguous(), V, None, 0, None, 0, None, self._input_flags,
                            attn_mask=am_dev)
        tile_chunks, seq_chunks = self._plan_passes(B, tiles, S)
        proj_all, rows_per_sample = None, 0
        if pixel_values is not None:
            v = cfg.mllm_config.vision_config
            C_l = cfg.mllm_config.text_config.hidden_size
            rows_per_sample = tiles * (v.num_patches + self.npt)
            proj_all = self._buf(("vit", "all"), "proj_all", (B * rows_per_sample, C_l))
            pvf = pv.reshape(B * tiles, *pv.shape[-3:])
            gmf = None if gm is None else gm.reshape(pvf.shape)
            t0 = 0
            for tc in tile_chunks:
                r0, r1 = t0 * (v.num_patches + self.npt), (t0 + tc) * (v.num_patches + self.npt)
                self.get_image_features(pvf[t0:t0 + tc], None if gmf is None else gmf[t0:t0 + tc], pooled=False,
                                        out=proj_all[r0:r1])
                t0 += tc
        b0 = 0
        for sc in seq_chunks:
            b1 = b0 + sc
            ids_c = input_ids[b0:b1]
            if pixel_values is not None:
                ar_c = None if aspect_ratios is None else aspect_ratios[b0:b1]
                embeds = self.build_inputs_embeds(ids_c, None, bboxes[b0:b1], ar_c, tiles, validate,
                                                  video_frame_tokens if feature_replay_video else None,
                                                  proj=proj_all[b0 * rows_per_sample:b1 * rows_per_sample])
            else:
                embeds = self._buf(("emb", b1 - b0, S), "embeds", (b1 - b0, S, cfg.mllm_config.text_config.hidden_size))
                ops.embed_assemble(ids_c.to(self.device, torch.int64).contiguous(), None, self.E, None, embeds, 0)
            last = self._prefill(embeds, st, Smax, b0)
            lg = self._head(last, b1 - b0, out_tokens[b0:b1], st, cur=st["cur"][b0:b1], finished=st["finished"][b0:b1], row0=b0)
            if forced_tokens is not None:
                st["cur"][b0:b1].copy_(forced_tokens[b0:b1, 0])
            if return_logits:
                first_logits.append(lg[:, :V].float().clone())
            b0 = b1
        all_logits = []
        if return_logits:
            all_logits.append(torch.cat(first_logits, 0))
        # counters after prefill: pos = S (position of the next token), kv_len = S+1 (incl. it), step = 1
        # (fills take their value as a kernel argument: no host-to-device copy, nothing the host waits for)
        c = st["counters"]
        c[0:1].fill_(S)
        c[1:2].fill_(S + 1)
        c[2:3].fill_(1)
        c[3:4].zero_()
        eos, eos_first = set(eos_list), (eos_list[0] if eos_list else None)
        return _PendingGeneration(eos_on_device=eos_on_device, st=st, skey=skey, B=B, Smax=Smax, V=V, out_tokens=out_tokens, max_new_tokens=max_new_tokens,
                                  eos=eos, eos_first=eos_first, pad_token_id=pad_token_id, return_logits=return_logits,
                                  all_logits=all_logits, forced_tokens=forced_tokens, use_graph=use_graph, validate=validate,
                                  sync_every=sync_every, input_flags=self._input_flags)

    @_on_model_device
    @torch.no_grad()
    def generate_finish(self, pend: "_PendingGeneration") -> GenerateOutput:
        """The decode loop of a :meth:`generate_begin` (current stream; the caller orders it behind the prompt phase)."""
        st, skey, B, Smax, V, out_tokens = pend.st, pend.skey, pend.B, pend.Smax, pend.V, pend.out_tokens
        max_new_tokens, eos, eos_first, pad_token_id = pend.max_new_tokens, pend.eos, pend.eos_first, pend.pad_token_id
        return_logits, all_logits, forced_tokens = pend.return_logits, pend.all_logits, pend.forced_tokens
        use_graph, validate, sync_every = pend.use_graph, pend.validate, pend.sync_every
        self._input_flags = pend.input_flags
        # tensors the prompt phase allocated (possibly on another stream: GenerationPipeline) and this loop reads: the caching
        # allocator must not hand their memory to that stream's next allocation before this stream is done with them
        cur_stream = torch.cuda.current_stream(self.device)
        for t_ in list(all_logits) + [forced_tokens, pend.input_flags]:
            if t_ is not None:
                t_.record_stream(cur_stream)
        graph = None
        if use_graph and max_new_tokens > 1:
            graph, graph_logits = self._decode_graph(st, B, Smax, out_tokens, skey)
        n_done = 1
        finished_at = [None] * B
        if ops.KERNEL_TIMERS is not None:       # bench.py's per-kernel timers: the decode steps' launches are priced separately
            ops.KERNEL_PHASE = "decode:"
        while n_done < max_new_tokens:
            if graph is not None:
                graph.replay()
                lg = graph_logits            # the buffer the captured step writes: copied out stream-ordered below
            else:
                lg = self._decode_step(st, B, Smax, out_tokens)
            if return_logits:
                all_logits.append(lg[:, :V].float().clone())
            n_done += 1
            if forced_tokens is not None and n_done - 1 < forced_tokens.shape[1]:
                st["cur"].copy_(forced_tokens[:, n_done - 1])
            if eos and (n_done % sync_every == 0 or n_done == max_new_tokens):
                # every row done? ONE int off the device (the argmax kernel latches each row's first eos step and counts the
                # latched rows); a list longer than MAX_EOS_IDS keeps the host scan of the token matrix
                fin = int(st["done_count"].item()) >= B if pend.eos_on_device else \
                    self._all_finished(out_tokens, n_done, eos, finished_at)
                if not validate:
                    self._raise_on_input_flags()        # the poll above synchronised already
                if fin:
                    break
        ops.KERNEL_PHASE = ""
        seq = out_tokens[:, :n_done].clone()
        if eos:
            if pend.eos_on_device:      # finished[b] = the column of row b's first eos (-1: none within n_done tokens)
                finished_at = [(c + 1 if 0 <= c < n_done else None) for c in st["finished"].tolist()]
            else:
                self._all_finished(out_tokens, n_done, eos, finished_at)
            cut = max((f if f is not None else n_done) for f in finished_at)
            if cut < n_done or any(f is not None and f < cut for f in finished_at):
                pad = pad_token_id if pad_token_id is not None else eos_first      # HF: pad defaults to eos_token_id[0]
                # rows that finished early hold `pad` behind their eos, as HF's loop writes them; the batch ends at the LAST row's eos
                cols = torch.arange(cut, device=self.device)[None, :]
                ends = torch.tensor([(f if f is not None else cut) for f in finished_at], dtype=torch.int64, device=self.device)
                seq = torch.where(cols < ends[:, None], seq[:, :cut], torch.full_like(seq[:, :cut], int(pad)))
        return GenerateOutput(sequences=seq, logits=torch.stack(all_logits, 1) if return_logits else None,
                              input_flags=None if validate else self._input_flags)

    def _raise_on_input_flags(self):
        f = int(self._input_flags.item())
        if f:
            raise ValueError("generate(validate=False): " + describe_input_flags(f))

    @staticmethod
    def _all_finished(out_tokens, n_done, eos, finished_at) -> bool:
        host = out_tokens[:, :n_done].tolist()
        for b, row in enumerate(host):
            if finished_at[b] is None:
                for j, tok in enumerate(row):
                    if tok in eos:
                        finished_at[b] = j + 1
                        break
        return all(f is not None for f in finished_at)

    def _decode_graph(self, st, B, Smax, out_tokens, skey):
        """(graph, logits buffer of the captured step). out_tokens is a [:, :n] slice of the state's [B, Smax] buffer:
        pointer and row stride depend on the state only, so one graph serves every max_new_tokens of the bucket."""
        gkey = (skey, out_tokens.data_ptr(), out_tokens.stride(0), bool(st.get("sampling")))
        g = self._graphs.get(gkey)
        if g is None:
            # what the warm-up step writes and the loop reads: counters, the current tokens, the stopping-criterion latches
            names = ("counters", "cur", "finished", "done_count")
            saved = {n: st[n].clone() for n in names}

            def restore():
                for n in names:
                    st[n].copy_(saved[n])
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):                       # warm-up launch outside capture (lazy module load)
                self._decode_step(st, B, Smax, out_tokens)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize(self.device)
            restore()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=s):
                logits = self._decode_step(st, B, Smax, out_tokens)
            restore()                                        # capture does not execute; keep state explicit
            g = self._graphs[gkey] = (g, logits)
        return g
