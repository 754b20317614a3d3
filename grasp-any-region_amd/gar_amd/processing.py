"""Host-side input contract of the hot path: tiling image processor, mask (visual prompt) branch,
`<|image|>` expansion, chat template and a tokenizer.

What it mirrors (behaviour, not code):
  * canvas search / thumb+tile split / rescale+normalize:
    projects/grasp_any_region/models/modeling/image_processing_perception_lm_fast.py:95-372
  * `<|image|>` -> (tile//patch//pool)^2 * tiles placeholder expansion:
    projects/grasp_any_region/models/modeling/processing_perception_lm.py:200-220
  * mask branch = the same pipeline with NEAREST resampling on the id-matrix image:
    projects/grasp_any_region/datasets/GraspAnyRegion_Dataset.py:123-128,686-699
  * processor call signature used by callers: evaluation/eval_dataset.py:122-139

The hub tokenizer / chat-template files ship with the released weights and are not available here, so
``StubTokenizer`` provides the fixed Llama-3 special ids with a reversible
byte-level text encoding; ``GARProcessor.from_pretrained`` uses the real tokenizer when a checkpoint
directory holds one.
"""
from __future__ import annotations

import math
import re
from functools import reduce
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image


# ------------------------------------------------------------------------------------------------
# canvas selection (image_processing_perception_lm_fast.py:95-252)
# ------------------------------------------------------------------------------------------------
def _factors(n: int):
    return set(reduce(list.__add__, ([i, n // i] for i in range(1, int(n ** 0.5) + 1) if n % i == 0)))


def find_supported_aspect_ratios(max_num_tiles: int) -> Dict[float, List[Tuple[int, int]]]:
    """Insertion order matters: ties in ``fit_image_to_canvas`` go to the later entry (:105-133)."""
    asp: Dict[float, List[Tuple[int, int]]] = {}
    for chunk in range(max_num_tiles, 0, -1):
        for x in sorted(_factors(chunk)):
            ratio = (x, chunk // x)
            asp.setdefault(ratio[0] / ratio[1], []).append(ratio)
    return asp


def _image_wh_in_canvas(iw: int, ih: int, tw: int, th: int):
    scale = iw / ih
    r = min(tw / iw, th / ih)
    if scale > 1.0:
        new_w = r * iw
        new_h = math.floor(new_w / scale)
    else:
        new_h = r * ih
        new_w = math.floor(new_h * scale)
    return new_w, new_h


def fit_image_to_canvas(iw: int, ih: int, tile: int, max_num_tiles: int) -> Optional[Tuple[int, int]]:
    best, best_wh = None, None
    scale = iw / ih
    arrangements = [it for sub in find_supported_aspect_ratios(max_num_tiles).values() for it in sub]
    for n_w, n_h in arrangements:
        cw, ch = n_w * tile, n_h * tile
        if cw >= iw and ch >= ih:
            wh = _image_wh_in_canvas(iw, ih, cw, ch)
            if best is None:
                best, best_wh = (n_w, n_h), wh
            elif (scale < 1.0 and wh[0] >= best_wh[0]) or (scale >= 1.0 and wh[1] >= best_wh[1]):
                best, best_wh = (n_w, n_h), wh
    return best


def find_closest_aspect_ratio(iw: int, ih: int, max_num_tiles: int) -> Tuple[int, int]:
    target = iw / ih
    asp = find_supported_aspect_ratios(max_num_tiles)
    if target >= 1:
        k = min([k for k in asp if k <= target], key=lambda x: abs(x - target))
        return max(asp[k], key=lambda x: x[0])
    k = min([k for k in asp if k > target], key=lambda x: abs(1 / x - 1 / target))
    return max(asp[k], key=lambda x: x[1])


def select_canvas(iw: int, ih: int, tile: int, max_num_tiles: int) -> Tuple[int, int]:
    """(tiles_w, tiles_h) exactly as ``resize`` picks it (:268-286)."""
    if max_num_tiles <= 1:
        return (1, 1)
    c = fit_image_to_canvas(iw, ih, tile, max_num_tiles)
    return c if c is not None else find_closest_aspect_ratio(iw, ih, max_num_tiles)


def split_tiles(img: torch.Tensor, ncw: int, nch: int) -> torch.Tensor:
    """[B,C,H,W] -> [B, ncw*nch, C, H/nch, W/ncw], tile index = h_idx*ncw + w_idx (:254-266)."""
    b, c, h, w = img.shape
    x = img.view(b, c, nch, h // nch, ncw, w // ncw).permute(0, 2, 4, 1, 3, 5).contiguous()
    return x.view(b, ncw * nch, c, h // nch, w // ncw)


def _resize_u8(img_u8: torch.Tensor, size_hw, resample: str) -> torch.Tensor:
    """uint8 [C,H,W] -> uint8 [C,h,w]; what torchvision's tensor ``F.resize`` does for uint8 input:
    float32 bicubic with antialias, round, clamp; NEAREST is index selection."""
    x = img_u8.unsqueeze(0).to(torch.float32)
    if resample == "nearest":
        y = F.interpolate(x, size=size_hw, mode="nearest")
    else:
        y = F.interpolate(x, size=size_hw, mode="bicubic", align_corners=False, antialias=True)
        y = y.round().clamp_(0, 255)
    return y.squeeze(0).to(torch.uint8)


class GARImageProcessor:
    """thumb+tile preprocessing; mean=std=0.5 (:76-77), RGB conversion (:82)."""

    def __init__(self, tile_size: int = 448, max_num_tiles: int = 16, resample: str = "bicubic"):
        self.tile_size = tile_size
        self.max_num_tiles = max_num_tiles
        self.resample = resample
        self.image_mean = 0.5
        self.image_std = 0.5

    def rescale_and_normalize(self, x: torch.Tensor) -> torch.Tensor:
        """HF ``BaseImageProcessorFast.rescale_and_normalize`` with do_rescale and do_normalize (what
        image_processing_perception_lm_fast.py:350-358 calls): mean and std are multiplied by 1 / rescale_factor = 255 and
        the image is normalised ONCE in fp32, (x - 127.5) / 127.5 — not (x / 255 - 0.5) / 0.5, which differs by one ulp
        for some uint8 values."""
        mean = torch.tensor(self.image_mean, dtype=torch.float32) * (1.0 / (1.0 / 255.0))
        std = torch.tensor(self.image_std, dtype=torch.float32) * (1.0 / (1.0 / 255.0))
        return (x.to(torch.float32) - mean) / std

    def single_tile(self, image: Image.Image, resample: Optional[str] = None) -> torch.Tensor:
        """One frame -> one normalised tile [1,3,ts,ts] (the ``max_num_tiles=1`` branch of ``resize``, :268-286)."""
        resample = resample or self.resample
        rgb = np.asarray(image.convert("RGB"), dtype=np.uint8)
        x = torch.from_numpy(rgb.copy()).permute(2, 0, 1).contiguous()
        t = _resize_u8(x, (self.tile_size, self.tile_size), resample).to(torch.float32)
        return self.rescale_and_normalize(t).unsqueeze(0)

    def __call__(self, image: Image.Image, resample: Optional[str] = None):
        resample = resample or self.resample
        rgb = np.asarray(image.convert("RGB"), dtype=np.uint8)
        x = torch.from_numpy(rgb.copy()).permute(2, 0, 1).contiguous()        # [3,H,W] uint8
        h, w = x.shape[1:]
        ts = self.tile_size
        thumb = _resize_u8(x, (ts, ts), resample)
        n_w, n_h = select_canvas(w, h, ts, self.max_num_tiles)
        big = _resize_u8(x, (n_h * ts, n_w * ts), resample)
        tiles = split_tiles(big.unsqueeze(0), n_w, n_h)[0]                     # [n, 3, ts, ts]
        stacked = torch.cat([thumb.unsqueeze(0), tiles], dim=0).to(torch.float32)
        pix = self.rescale_and_normalize(stacked)
        return pix.unsqueeze(0), [n_w, n_h]                                    # [1, T+1, 3, ts, ts]


# ------------------------------------------------------------------------------------------------
# tokenizer
# ------------------------------------------------------------------------------------------------
LLAMA3_SPECIALS = {
    "<|begin_of_text|>": 128000, "<|end_of_text|>": 128001, "<|image|>": 128002, "<|video|>": 128003,
    "<|reserved_special_token_2|>": 128004, "<|reserved_special_token_3|>": 128005,
    "<|start_header_id|>": 128006, "<|end_header_id|>": 128007,
    "<|reserved_special_token_4|>": 128008, "<|eot_id|>": 128009,
    "<|reserved_special_token_5|>": 128010, "<|reserved_special_token_6|>": 128011,
    "<|reserved_special_token_7|>": 128012, "<|reserved_special_token_8|>": 128013,
    "<|reserved_special_token_9|>": 128014,
    "<Prompt0>": 128256, "<Prompt1>": 128257, "<Prompt2>": 128258, "<Prompt3>": 128259,
    "<Prompt4>": 128260, "<NO_Prompt>": 128261,
}

# the same roles at ids < 512 for GARConfig.tiny()
TINY_SPECIALS = {
    "<|begin_of_text|>": 296, "<|end_of_text|>": 297, "<|image|>": 300, "<|video|>": 301,
    "<|reserved_special_token_2|>": 304, "<|reserved_special_token_3|>": 305,
    "<|start_header_id|>": 306, "<|end_header_id|>": 307,
    "<|reserved_special_token_4|>": 308, "<|eot_id|>": 309,
    "<|reserved_special_token_5|>": 310, "<|reserved_special_token_6|>": 311,
    "<|reserved_special_token_7|>": 312, "<|reserved_special_token_8|>": 313, "<|reserved_special_token_9|>": 314,
    "<Prompt0>": 320, "<Prompt1>": 321, "<Prompt2>": 322, "<Prompt3>": 323, "<Prompt4>": 324,
    "<NO_Prompt>": 325,
}


class StubTokenizer:
    """Reversible byte-level tokenizer with the fixed special ids. Text bytes map to ids
    ``byte_offset + byte``; special tokens are matched greedily before byte encoding."""

    def __init__(self, specials: Dict[str, int] = None, byte_offset: int = 1000, prompt_base: int = None):
        self.specials = dict(specials or LLAMA3_SPECIALS)
        self.byte_offset = byte_offset
        self.inv = {v: k for k, v in self.specials.items()}
        self._re = re.compile("(" + "|".join(re.escape(k) for k in sorted(self.specials, key=len, reverse=True)) + ")")
        self.image_token = "<|image|>"
        self.video_token = "<|video|>"
        self.image_token_id = self.specials["<|image|>"]
        self.video_token_id = self.specials["<|video|>"]
        self.eos_token_id = self.specials["<|eot_id|>"]
        self.pad_token_id = self.specials["<|end_of_text|>"]
        self.bos_token_id = self.specials["<|begin_of_text|>"]
        # visual prompt ids are "token id - prompt_base" (eval_dataset.py:44-47 uses 128256)
        self.prompt_base = self.specials["<Prompt0>"] if prompt_base is None else prompt_base

    @classmethod
    def tiny(cls):
        return cls(TINY_SPECIALS, byte_offset=0)

    def convert_tokens_to_ids(self, token):
        if isinstance(token, (list, tuple)):
            return [self.convert_tokens_to_ids(t) for t in token]
        return self.specials[token]

    def encode(self, text: str) -> List[int]:
        ids: List[int] = []
        for piece in self._re.split(text):
            if not piece:
                continue
            if piece in self.specials:
                ids.append(self.specials[piece])
            else:
                ids.extend(self.byte_offset + b for b in piece.encode("utf-8"))
        return ids

    def __call__(self, texts: Sequence[str], **_):
        enc = [self.encode(t) for t in texts]
        return {"input_ids": enc, "attention_mask": [[1] * len(e) for e in enc]}

    def decode(self, ids, skip_special_tokens: bool = False) -> str:
        if torch.is_tensor(ids):
            ids = ids.tolist()
        out = bytearray()
        for i in ids:
            i = int(i)
            if i in self.inv:
                if not skip_special_tokens:
                    out.extend(self.inv[i].encode())
            elif self.byte_offset <= i < self.byte_offset + 256:
                out.append(i - self.byte_offset)
            else:
                if not skip_special_tokens:
                    out.extend(f"<{i}>".encode())
        return out.decode("utf-8", errors="replace")


class _HFTokenizerAdapter:
    """Thin adapter over a real HF tokenizer directory (only used when one exists on disk)."""

    def __init__(self, path: str):
        from transformers import AutoTokenizer
        self.tk = AutoTokenizer.from_pretrained(path)
        self.image_token = "<|image|>"
        self.video_token = "<|video|>"
        self.image_token_id = self.tk.convert_tokens_to_ids(self.image_token)
        self.video_token_id = self.tk.convert_tokens_to_ids(self.video_token)
        self.eos_token_id = self.tk.eos_token_id
        self.pad_token_id = self.tk.pad_token_id
        self.prompt_base = 128256

    def convert_tokens_to_ids(self, t):
        return self.tk.convert_tokens_to_ids(t)

    def __call__(self, texts, **kw):
        return self.tk(list(texts), add_special_tokens=False)

    def decode(self, ids, skip_special_tokens=False):
        return self.tk.decode(ids, skip_special_tokens=skip_special_tokens)


# ------------------------------------------------------------------------------------------------
# processor
# ------------------------------------------------------------------------------------------------
class GARProcessor:
    """``processor(text=[str], images=[PIL], visual_prompts=[PIL], return_tensors="pt")`` ->
    dict(pixel_values [1,T+1,3,ts,ts], mask_values [1,T+1,3,ts,ts], input_ids [1,S], attention_mask [1,S],
    aspect_ratio tensor([n_w, n_h]))   (eval_dataset.py:128-139)."""

    def __init__(self, tokenizer=None, tile_size: int = 448, max_num_tiles: int = 16, patch_size: int = 14,
                 pooling_ratio: int = 2, chat_template: str = None):
        self.tokenizer = tokenizer or StubTokenizer()
        self.chat_template = chat_template          # Jinja source from a checkpoint directory (None: the built-in Llama-3 / PLM layout)
        self.image_processor = GARImageProcessor(tile_size, max_num_tiles, "bicubic")
        self.patch_size = patch_size
        self.pooling_ratio = pooling_ratio

    @classmethod
    def from_config(cls, cfg, max_num_tiles: int = 16, tokenizer=None):
        v = cfg.mllm_config.vision_config
        if tokenizer is None:
            tokenizer = StubTokenizer() if cfg.mllm_config.image_token_id == 128002 else StubTokenizer.tiny()
        return cls(tokenizer, v.img_size, max_num_tiles, v.patch_size, cfg.mllm_config.projector_pooling_ratio)

    def use_gpu_preprocessing(self, device="cuda:0", dtype: torch.dtype = torch.bfloat16):
        """Swap the host image processor for the device one (``preprocess_gpu.GpuImageProcessor``): same contract,
        ``pixel_values`` / ``mask_values`` come back as device tensors in ``dtype``. Returns self."""
        from .preprocess_gpu import GpuImageProcessor
        ip = self.image_processor
        self.image_processor = GpuImageProcessor(ip.tile_size, ip.max_num_tiles, "bicubic", device, dtype)
        return self

    @classmethod
    def from_pretrained(cls, path: str, cfg=None, max_num_tiles: int = 16):
        import os
        tk = None
        if os.path.isdir(path) and any(os.path.exists(os.path.join(path, f))
                                      for f in ("tokenizer.json", "tokenizer.model")):
            tk = _HFTokenizerAdapter(path)
        if cfg is None:
            from .configuration_gar import GARConfig
            cj = os.path.join(path, "config.json")
            cfg = GARConfig.from_json_file(cj) if os.path.exists(cj) else GARConfig.gar_1b()
        proc = cls.from_config(cfg, max_num_tiles, tk)
        proc.chat_template = cls.read_chat_template(path)
        return proc

    @staticmethod
    def read_chat_template(path: str):
        """The chat template a hub snapshot carries, in transformers' own order of precedence (ProcessorMixin /
        PreTrainedTokenizerBase): chat_template.jinja, chat_template.json, processor_config.json, tokenizer_config.json.
        None when the directory has none (the built-in layout is used then)."""
        import json
        import os
        p = os.path.join(path, "chat_template.jinja")
        if os.path.isfile(p):
            return open(p, encoding="utf-8").read()
        for name in ("chat_template.json", "processor_config.json", "tokenizer_config.json"):
            p = os.path.join(path, name)
            if os.path.isfile(p):
                tpl = json.load(open(p, encoding="utf-8")).get("chat_template")
                if isinstance(tpl, str) and tpl:
                    return tpl
        return None

    def _compiled_chat_template(self):
        """The checkpoint's template compiled once. transformers' own (private) compiler when this version has it — it adds
        the ``raise_exception`` / ``strftime_now`` globals and the JSON filter HF templates may use —, else the same Jinja
        sandbox built here (no ImportError on a transformers without that symbol)."""
        cached = getattr(self, "_chat_template_compiled", None)
        if cached is not None and cached[0] == self.chat_template:
            return cached[1]
        try:
            from transformers.utils.chat_template_utils import _compile_jinja_template
            tpl = _compile_jinja_template(self.chat_template)
        except ImportError:
            import datetime
            import json

            import jinja2
            from jinja2.sandbox import ImmutableSandboxedEnvironment

            def raise_exception(message):
                raise jinja2.exceptions.TemplateError(message)

            env = ImmutableSandboxedEnvironment(trim_blocks=True, lstrip_blocks=True)
            env.filters["tojson"] = lambda x, **kw: json.dumps(x, ensure_ascii=False, **{k: v for k, v in kw.items() if k in ("indent", "sort_keys")})
            env.globals["raise_exception"] = raise_exception
            env.globals["strftime_now"] = lambda fmt: datetime.datetime.now().strftime(fmt)
            tpl = env.from_string(self.chat_template)
        self._chat_template_compiled = (self.chat_template, tpl)
        return tpl

    def apply_chat_template(self, messages, add_generation_prompt: bool = True, tokenize: bool = False) -> str:
        """The checkpoint's own chat template when its directory had one (rendered by transformers' Jinja environment, as
        ``processor.apply_chat_template`` of the reference does, evaluation/eval_dataset.py:122); otherwise the Llama-3 /
        PLM layout it encodes: an image item becomes one ``<|image|>`` ahead of the text."""
        assert not tokenize
        if self.chat_template:
            tk = getattr(self.tokenizer, "tk", None)
            special = dict(getattr(tk, "special_tokens_map", None) or {})
            special.setdefault("bos_token", "<|begin_of_text|>")
            special.setdefault("eos_token", "<|eot_id|>")
            return self._compiled_chat_template().render(messages=messages, add_generation_prompt=add_generation_prompt,
                                                         **special)
        s = "<|begin_of_text|>"
        for m in messages:
            s += f"<|start_header_id|>{m['role']}<|end_header_id|>\n\n"
            content = m["content"]
            if isinstance(content, str):
                s += content
            else:
                for item in content:
                    if item["type"] == "image":
                        s += self.tokenizer.image_token
                    elif item["type"] == "text":
                        s += item["text"]
            s += "<|eot_id|>"
        if add_generation_prompt:
            s += "<|start_header_id|>assistant<|end_header_id|>\n\n"
        return s

    def num_image_tokens(self, n_tiles: int) -> int:
        side = self.image_processor.tile_size // self.patch_size // self.pooling_ratio
        return side * side * n_tiles

    def __call__(self, text, images=None, visual_prompts=None, return_tensors="pt"):
        if text is None:
            raise ValueError("You have to specify at least `text` input.")
        if isinstance(text, str):
            text = [text]
        out = {}
        pix_list, ar = [], [1, 1]
        if images is not None:
            for im in images:
                pix, ar = self.image_processor(im, "bicubic")
                pix_list.append(pix)
            out["pixel_values"] = torch.cat(pix_list, dim=0)
            out["aspect_ratio"] = torch.tensor(ar, dtype=torch.int64)
        if visual_prompts is not None:
            out["mask_values"] = torch.cat([self.image_processor(vp, "nearest")[0] for vp in visual_prompts], dim=0)
        # expand each <|image|> to its placeholder run (processing_perception_lm.py:200-220)
        it = iter(pix_list)
        prompts = []
        tok = self.tokenizer.image_token
        for sample in text:
            n = sample.count(tok)
            if n:
                parts = sample.split(tok)
                s = ""
                for i in range(n):
                    s += parts[i] + tok * self.num_image_tokens(next(it).shape[1])
                sample = s + parts[-1]
            prompts.append(sample)
        enc = self.tokenizer(prompts)
        out["input_ids"] = torch.tensor(enc["input_ids"], dtype=torch.int64)
        out["attention_mask"] = torch.tensor(enc["attention_mask"], dtype=torch.int64)
        return out
