"""Configuration contract of the GAR region-captioning hot path.

Mirrors the attribute surface the reference callers and model read:

* ``GARConfig(mllm_config, prompt_numbers, crop_tokens_ids)`` with the derived
  ``kernel_size`` and ``mask_path_embedding_out_channels``
  (reference: projects/grasp_any_region/hf_models/configuration_gar.py:10-64)
* ``PerceptionLMConfig`` with ``vision_config.model_args``, ``text_config``,
  ``vision_use_cls_token``, ``projector_pooling_ratio``, ``image_token_id``,
  ``video_token_id``
  (reference: projects/grasp_any_region/models/modeling/configuration_perception_lm.py:26-86)

It is a plain-Python object (no transformers dependency) that accepts the same
nested dict a HF ``config.json`` of a GAR checkpoint holds, so the real
checkpoints' configs load through ``GARConfig.from_dict``.
Model dimensions of the two released sizes are the upstream Perception-LM ones
(SURVEY.md Appendix B).
"""
from __future__ import annotations

import copy
import json
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional


@dataclass
class VisionConfig:
    """timm PE ViT ("vit_pe_lang_*") arguments; ``model_args`` is the dict the
    reference reads (grasp_any_region.py:69-80, configuration_gar.py:40-53)."""

    model_args: Dict[str, Any] = field(default_factory=dict)
    architecture: str = "vit_pe_lang_large_patch14_448"
    num_features: int = 1024

    # ---- derived views used by kernels/oracle -------------------------------------------
    @property
    def embed_dim(self) -> int:
        return int(self.model_args["embed_dim"])

    @property
    def depth(self) -> int:
        return int(self.model_args["depth"])

    @property
    def num_heads(self) -> int:
        return int(self.model_args.get("num_heads", 16))

    @property
    def head_dim(self) -> int:
        return self.embed_dim // self.num_heads

    @property
    def mlp_dim(self) -> int:
        if "mlp_dim" in self.model_args:
            return int(self.model_args["mlp_dim"])
        return int(round(self.embed_dim * float(self.model_args.get("mlp_ratio", 4.0))))

    @property
    def img_size(self) -> int:
        s = self.model_args["img_size"]
        return int(s[0] if isinstance(s, (list, tuple)) else s)

    @property
    def grid(self) -> int:
        s = self.model_args["ref_feat_shape"]
        return int(s[0] if isinstance(s, (list, tuple)) else s)

    @property
    def patch_size(self) -> int:
        return self.img_size // self.grid

    @property
    def num_patches(self) -> int:
        return self.grid * self.grid

    @property
    def init_values(self) -> float:
        return float(self.model_args.get("init_values", 0.1))

    @property
    def ln_eps(self) -> float:
        return float(self.model_args.get("ln_eps", 1e-5))

    @property
    def rope_temperature(self) -> float:
        return float(self.model_args.get("rope_temperature", 10000.0))

    @property
    def rope_grid_offset(self) -> float:
        return float(self.model_args.get("rope_grid_offset", 1.0))

    @property
    def rope_grid_indexing(self) -> str:
        return str(self.model_args.get("rope_grid_indexing", "xy"))

    def to_dict(self):
        return {"model_args": copy.deepcopy(self.model_args), "architecture": self.architecture,
                "num_features": self.num_features}


@dataclass
class TextConfig:
    """HF ``LlamaConfig`` field names (text_config of PerceptionLMConfig)."""

    hidden_size: int = 2048
    num_hidden_layers: int = 16
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    head_dim: int = 64
    intermediate_size: int = 8192
    vocab_size: int = 128262
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0
    rope_scaling: Optional[Dict[str, Any]] = None
    tie_word_embeddings: bool = True
    max_position_embeddings: int = 131072

    def to_dict(self):
        return copy.deepcopy(self.__dict__)


@dataclass
class PerceptionLMConfig:
    vision_config: VisionConfig = field(default_factory=VisionConfig)
    text_config: TextConfig = field(default_factory=TextConfig)
    vision_use_cls_token: bool = True
    projector_pooling_ratio: int = 2
    image_token_id: int = 128002
    video_token_id: int = 128003

    def to_dict(self):
        return {
            "vision_config": self.vision_config.to_dict(),
            "text_config": self.text_config.to_dict(),
            "vision_use_cls_token": self.vision_use_cls_token,
            "projector_pooling_ratio": self.projector_pooling_ratio,
            "image_token_id": self.image_token_id,
            "video_token_id": self.video_token_id,
        }

    @classmethod
    def from_dict(cls, d: Dict[str, Any]) -> "PerceptionLMConfig":
        v = d.get("vision_config", {})
        t = dict(d.get("text_config", {}))
        tfields = TextConfig.__dataclass_fields__.keys()
        tcfg = TextConfig(**{k: t[k] for k in t if k in tfields})
        if "head_dim" not in t or t.get("head_dim") is None:
            tcfg.head_dim = tcfg.hidden_size // tcfg.num_attention_heads
        margs = dict(v.get("model_args", {}))
        vcfg = VisionConfig(model_args=margs, architecture=v.get("architecture", ""),
                            num_features=int(v.get("num_features", margs.get("embed_dim", 1024))))
        return cls(
            vision_config=vcfg,
            text_config=tcfg,
            vision_use_cls_token=bool(d.get("vision_use_cls_token", True)),
            projector_pooling_ratio=int(d.get("projector_pooling_ratio", 2)),
            image_token_id=int(d.get("image_token_id", 128002)),
            video_token_id=int(d.get("video_token_id", 128003)),
        )


class GARConfig:
    """Reference: hf_models/configuration_gar.py:10-64 (same attribute names)."""

    model_type = "GAR"

    def __init__(self, mllm_config=None, prompt_numbers: int = 5,
                 crop_tokens_ids: Optional[List[int]] = None, **kwargs):
        if crop_tokens_ids is None:
            crop_tokens_ids = [128004, 128005, 128008, 128010, 128011]
        if mllm_config is None:
            mllm_config = _plm_1b_dict()
        if isinstance(mllm_config, dict):
            mllm_config = PerceptionLMConfig.from_dict(mllm_config)
        self.mllm_config: PerceptionLMConfig = mllm_config
        self.prompt_numbers = int(prompt_numbers)
        self.crop_tokens_ids = [int(t) for t in crop_tokens_ids]
        # same check as the reference (configuration_gar.py:36-38)
        assert len(self.crop_tokens_ids) == self.prompt_numbers, (
            f"{self.crop_tokens_ids} crop_tokens_ids length should be {self.prompt_numbers}")
        v = self.mllm_config.vision_config
        self.patch_size_h = v.patch_size
        self.patch_size_w = v.patch_size
        self.kernel_size = [self.patch_size_h, self.patch_size_w]
        self.mask_path_embedding_out_channels = v.num_features
        self.extra = kwargs

    # ---- geometry the reference hard-codes as 16 / 256 / 28 (modeling_gar.py:350-367) ------
    @property
    def pooled_side(self) -> int:
        v = self.mllm_config.vision_config
        return v.grid // self.mllm_config.projector_pooling_ratio

    @property
    def tokens_per_tile(self) -> int:
        return self.pooled_side * self.pooled_side

    @property
    def feat_stride(self) -> int:
        """pixels of the tiled canvas per pooled feature cell (= 28 upstream)."""
        return self.mllm_config.vision_config.patch_size * self.mllm_config.projector_pooling_ratio

    def to_dict(self):
        return {"model_type": self.model_type, "mllm_config": self.mllm_config.to_dict(),
                "prompt_numbers": self.prompt_numbers, "crop_tokens_ids": list(self.crop_tokens_ids)}

    @classmethod
    def from_dict(cls, d):
        return cls(mllm_config=d.get("mllm_config"), prompt_numbers=d.get("prompt_numbers", 5),
                   crop_tokens_ids=d.get("crop_tokens_ids"))

    @classmethod
    def from_json_file(cls, path):
        with open(path) as f:
            return cls.from_dict(json.load(f))

    # ---- named sizes --------------------------------------------------------------------------
    @classmethod
    def gar_1b(cls, **over):
        return cls(mllm_config=_apply(_plm_1b_dict(), over))

    @classmethod
    def gar_8b(cls, **over):
        return cls(mllm_config=_apply(_plm_8b_dict(), over))

    @classmethod
    def tiny(cls, **over):
        """Small config for CPU-speed parity tests: same structure (cls token, GQA, tied head,
        llama3 rope scaling, 2x2 pooling), every spatial constant scaled down
        (112 px tiles -> 8x8 patches -> 4x4 pooled tokens -> 16 crop tokens)."""
        d = {
            "vision_config": {"architecture": "vit_pe_lang_tiny_patch14_112", "num_features": 128,
                              "model_args": {"embed_dim": 128, "depth": 2, "num_heads": 2, "mlp_dim": 256,
                                             "img_size": [112, 112], "ref_feat_shape": [8, 8],
                                             "init_values": 0.1, "global_pool": "",
                                             "use_post_transformer_norm": False}},
            "text_config": {"hidden_size": 128, "num_hidden_layers": 2, "num_attention_heads": 4,
                            "num_key_value_heads": 2, "head_dim": 64, "intermediate_size": 256,
                            "vocab_size": 512, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
                            "rope_scaling": {"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                             "original_max_position_embeddings": 8192, "rope_type": "llama3"},
                            "tie_word_embeddings": True},
            "vision_use_cls_token": True, "projector_pooling_ratio": 2,
            "image_token_id": 300, "video_token_id": 301,
        }
        d = _apply(d, over)
        d["vision_config"]["num_features"] = d["vision_config"]["model_args"]["embed_dim"]
        return cls(mllm_config=d, prompt_numbers=5, crop_tokens_ids=[304, 305, 308, 310, 311])


def _apply(d, over):
    """over: {"vision.depth": 2, "text.num_hidden_layers": 2, ...} shallow overrides."""
    for k, val in over.items():
        sec, _, key = k.partition(".")
        if sec == "vision":
            d["vision_config"]["model_args"][key] = val
        elif sec == "text":
            d["text_config"][key] = val
        else:
            d[k] = val
    return d


_LLAMA3_SCALING_1B = {"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                      "original_max_position_embeddings": 8192, "rope_type": "llama3"}
_LLAMA3_SCALING_8B = {"factor": 8.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                      "original_max_position_embeddings": 8192, "rope_type": "llama3"}


def _plm_1b_dict():
    # facebook/Perception-LM-1B (reference configs/gar_1b.py:24); dims: SURVEY.md Appendix B
    return {
        "vision_config": {"architecture": "vit_pe_lang_large_patch14_448", "num_features": 1024,
                          "model_args": {"embed_dim": 1024, "depth": 23, "num_heads": 16, "mlp_dim": 4096,
                                         "img_size": [448, 448], "ref_feat_shape": [32, 32],
                                         "init_values": 0.1, "global_pool": "",
                                         "use_post_transformer_norm": False}},
        "text_config": {"hidden_size": 2048, "num_hidden_layers": 16, "num_attention_heads": 32,
                        "num_key_value_heads": 8, "head_dim": 64, "intermediate_size": 8192,
                        "vocab_size": 128262, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
                        "rope_scaling": dict(_LLAMA3_SCALING_1B), "tie_word_embeddings": True},
        "vision_use_cls_token": True, "projector_pooling_ratio": 2,
        "image_token_id": 128002, "video_token_id": 128003,
    }


def _plm_8b_dict():
    # facebook/Perception-LM-8B (reference configs/gar_8b.py:24)
    return {
        "vision_config": {"architecture": "vit_pe_lang_gigantic_patch14_448", "num_features": 1536,
                          "model_args": {"embed_dim": 1536, "depth": 47, "num_heads": 16, "mlp_dim": 8960,
                                         "img_size": [448, 448], "ref_feat_shape": [32, 32],
                                         "init_values": 0.1, "global_pool": "",
                                         "use_post_transformer_norm": False}},
        "text_config": {"hidden_size": 4096, "num_hidden_layers": 32, "num_attention_heads": 32,
                        "num_key_value_heads": 8, "head_dim": 128, "intermediate_size": 14336,
                        "vocab_size": 128262, "rms_norm_eps": 1e-5, "rope_theta": 500000.0,
                        "rope_scaling": dict(_LLAMA3_SCALING_8B), "tie_word_embeddings": False},
        "vision_use_cls_token": False, "projector_pooling_ratio": 2,
        "image_token_id": 128002, "video_token_id": 128003,
    }
