"""Sample builders: image + mask(s) (+ question) -> the kwargs dict of ``GARModel.generate``.

Own counterpart of the reference's ``evaluation/eval_dataset.py`` (``SingleRegionCaptionDataset`` :18-149,
``MultiRegionDataset`` :152-313): same constructor arguments, same output keys/shapes/dtypes, same prompt
strings, and the reference quirks parity depends on:

  * bbox = (xmin/W, ymin/H, xmax/W, ymax/H), inclusive max, no +1                 (:77-85)
  * id-matrix: int16 ``-1`` init (what ``-1*np.ones(uint8)`` yields under the pinned numpy 1.26),
    mask pixels -> prompt id, rest -> NO_Prompt id                                 (:63-72)
  * MultiRegion: every region's bbox comes from the LAST mask (stale ``mask_id``)  (:201,241)
  * MultiRegion: prompt order = iteration order of ``set(re.findall(...))``        (:207); pass
    ``prompt_order`` to pin it (goldens record the order they used)
"""
from __future__ import annotations

import re
from copy import deepcopy
from typing import List, Optional

import numpy as np
import torch
from PIL import Image

DEFAULT_VP_TOKENS = ["<Prompt0>", "<Prompt1>", "<Prompt2>", "<Prompt3>", "<Prompt4>", "<NO_Prompt>"]
CROP_REPEAT_DEFAULT = 256


def _default_device():
    return "cuda" if torch.cuda.is_available() else "cpu"


class _Base:
    def _finish(self, data_dict, qs):
        image = data_dict["image"]
        messages = [{"role": "user", "content": [{"type": "image", "image": image}, {"type": "text", "text": qs}]}]
        raw_prompt = self.processor.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
        mi = self.processor(text=[raw_prompt], images=[image], visual_prompts=[data_dict["visual_prompt"]],
                            return_tensors="pt")
        dev = self.device
        input_ids = mi["input_ids"].squeeze(0)
        attention_mask = mi["attention_mask"].squeeze(0)
        return dict(
            input_ids=input_ids.to(dev).unsqueeze(0),
            attention_mask=attention_mask.to(dev).to(self.data_dtype).unsqueeze(0),
            pixel_values=mi["pixel_values"].to(dev).to(self.data_dtype).flatten(0, 1),
            global_mask_values=mi["mask_values"].to(dev).to(self.data_dtype).squeeze(),
            bboxes=[data_dict["bboxes"]],
            aspect_ratios=mi["aspect_ratio"].unsqueeze(0).to(dev),
        )

    def _crop_repeat(self):
        n = getattr(self.processor, "num_image_tokens", None)
        return n(1) if n else CROP_REPEAT_DEFAULT


class SingleRegionCaptionDataset(_Base):
    def __init__(self, image, mask, processor, prompt_token="<Prompt1>", prompt_number=5,
                 visual_prompt_tokens=None, data_dtype=torch.bfloat16, device=None, **kwargs):
        self.processor = processor
        self.prompt_token = prompt_token
        self.prompt_number = prompt_number
        self.special_tokens = visual_prompt_tokens or DEFAULT_VP_TOKENS
        base = getattr(processor.tokenizer, "prompt_base", 128256)
        self.visual_prompt_ids = {t: processor.tokenizer.convert_tokens_to_ids(t) - base for t in self.special_tokens}
        self.image = image
        self.mask = mask
        self.data_dtype = data_dtype
        self.device = device or _default_device()

    def __len__(self):
        return 1

    def _parse_annotations(self):
        image, mask = self.image, self.mask
        mask_np = np.asarray(mask).astype(np.uint8)
        filled = -1 * np.ones((image.height, image.width), dtype=np.int16)
        prompt_id = self.visual_prompt_ids.get(self.prompt_token, self.visual_prompt_ids["<NO_Prompt>"])
        assert prompt_id < 16, f"prompt_id should be less than {16}, got {prompt_id}"
        filled[(filled == -1) & mask_np.astype(bool)] = prompt_id
        filled[filled == -1] = self.visual_prompt_ids["<NO_Prompt>"]
        prompt_idx = int(re.match(r"<Prompt(\d+)>", self.prompt_token).group(1))
        nz = np.argwhere(mask_np)
        y_min, x_min = nz.min(axis=0)
        y_max, x_max = nz.max(axis=0)
        bbox = (x_min / image.width, y_min / image.height, x_max / image.width, y_max / image.height)
        key = str(self.processor.tokenizer.convert_tokens_to_ids(f"<|reserved_special_token_{prompt_idx + 2}|>"))
        return {"image": image, "visual_prompt": Image.fromarray(filled.astype(np.int32)), "bboxes": {key: bbox}}

    def __getitem__(self, index):
        d = deepcopy(self._parse_annotations())
        prompt_idx = int(re.match(r"<Prompt(\d+)>", self.prompt_token).group(1))
        crop = f"<|reserved_special_token_{prompt_idx + 2}|>"
        qs = (f"There are some objects I am curious about: {self.prompt_token};\n{self.prompt_token}: "
              f"{crop * self._crop_repeat()}Describe this masked region in detail.")
        return self._finish(d, qs)


class MultiRegionDataset(_Base):
    def __init__(self, image, masks, question_str, processor, prompt_token="<Prompt1>", prompt_number=5,
                 visual_prompt_tokens=None, data_dtype=torch.bfloat16, device=None,
                 prompt_order: Optional[List[str]] = None, **kwargs):
        self.processor = processor
        self.prompt_token = prompt_token
        self.prompt_number = prompt_number
        self.special_tokens = visual_prompt_tokens or DEFAULT_VP_TOKENS
        base = getattr(processor.tokenizer, "prompt_base", 128256)
        self.visual_prompt_ids = {t: processor.tokenizer.convert_tokens_to_ids(t) - base for t in self.special_tokens}
        self.image = image
        self.masks = masks
        self.question_str = question_str
        self.data_dtype = data_dtype
        self.device = device or _default_device()
        self.prompt_order = prompt_order

    def __len__(self):
        return 1

    def _parse_annotations(self):
        image = self.image
        masks = list(self.masks)
        masks_np = [np.array(m).astype(np.uint8) for m in masks]
        mask_id = 0
        for mask_id, mask in enumerate(masks_np):                     # leaves mask_id = last index (:201)
            if image.width != mask.shape[1] or image.height != mask.shape[0]:
                m = Image.fromarray(mask).resize(image.size, Image.NEAREST)
                masks[mask_id] = np.array(m)
                masks_np[mask_id] = np.array(m).astype(np.uint8)
        found = set(re.findall(r"<Prompt\d+>", self.question_str))
        assert len(found) == len(masks)
        order = list(self.prompt_order) if self.prompt_order is not None else list(found)
        assert set(order) == found
        rep = self._crop_repeat()
        objects_desc = "There are some objects I am curious about: "
        sub = ""
        for p in order:
            objects_desc += f"{p}; "
            k = int(re.match(r"<Prompt(\d+)>", p).group(1))
            sub += f"{p}: " + f"<|reserved_special_token_{k + 2}|>" * rep + "\n"
        prompt = objects_desc + "\n" + sub + "\n" + self.question_str
        filled = -1 * np.ones((image.height, image.width), dtype=np.int16)
        bboxes = {}
        for p in order:
            k = int(re.match(r"<Prompt(\d+)>", p).group(1))
            mask = np.asarray(masks[k])
            prompt_id = self.visual_prompt_ids.get(p, self.visual_prompt_ids["<NO_Prompt>"])
            assert prompt_id < self.prompt_number + 1
            filled[(filled == -1) & mask.astype(bool)] = prompt_id      # first writer wins (:238-239)
            nz = np.argwhere(masks_np[mask_id])                         # stale mask_id: the LAST mask (:241)
            y_min, x_min = nz.min(axis=0)
            y_max, x_max = nz.max(axis=0)
            bbox = (x_min / image.width, y_min / image.height, x_max / image.width, y_max / image.height)
            key = str(self.processor.tokenizer.convert_tokens_to_ids(f"<|reserved_special_token_{k + 2}|>"))
            bboxes[key] = bbox
        filled[filled == -1] = self.visual_prompt_ids["<NO_Prompt>"]
        return {"image": image, "visual_prompt": Image.fromarray(filled.astype(np.int32)), "bboxes": bboxes,
                "prompt": prompt}

    def __getitem__(self, index):
        d = self._parse_annotations()
        return self._finish(d, d["prompt"])


class VideoRegionCaptionDataset:
    """Sample builder for the video replay path (A13, modeling_perception_lm.py:765-852). The reference ships no caller
    for that path; this builder follows its contract: F frames, each resized to ONE tile (no
    thumbnail), one binary mask per frame drawn with ``prompt_token``'s id, frame f's 256 placeholders are
    ``<|reserved_special_token_{2+f}|>`` and its bbox is keyed by that token's id. The returned dict adds
    ``feature_replay_video=True`` and ``video_frame_tokens`` for ``GARModel.generate``."""

    def __init__(self, frames, masks, processor, prompt_token="<Prompt1>", data_dtype=torch.bfloat16, device=None,
                 question="Describe this masked region in the video in detail.", **kwargs):
        assert len(frames) == len(masks) and 1 <= len(frames) <= 8
        self.frames, self.masks, self.processor = frames, masks, processor
        self.prompt_token, self.question = prompt_token, question
        self.data_dtype = data_dtype
        self.device = device or _default_device()
        base = getattr(processor.tokenizer, "prompt_base", 128256)
        tk = processor.tokenizer
        self.prompt_id = tk.convert_tokens_to_ids(prompt_token) - base
        self.no_prompt_id = tk.convert_tokens_to_ids("<NO_Prompt>") - base

    def __len__(self):
        return 1

    def __getitem__(self, index):
        tk, ip = self.processor.tokenizer, self.processor.image_processor
        rep = self.processor.num_image_tokens(1)
        pix, msk, bboxes, frame_tokens = [], [], {}, []
        body = ""
        for f, (frame, mask) in enumerate(zip(self.frames, self.masks)):
            mask_np = np.asarray(mask).astype(bool)
            filled = np.full((frame.height, frame.width), self.no_prompt_id, dtype=np.int32)
            filled[mask_np] = self.prompt_id
            nz = np.argwhere(mask_np)
            (y_min, x_min), (y_max, x_max) = nz.min(axis=0), nz.max(axis=0)
            tok = f"<|reserved_special_token_{f + 2}|>"
            tid = tk.convert_tokens_to_ids(tok)
            frame_tokens.append(tid)
            bboxes[str(tid)] = (x_min / frame.width, y_min / frame.height, x_max / frame.width, y_max / frame.height)
            pix.append(ip.single_tile(frame, "bicubic"))
            msk.append(ip.single_tile(Image.fromarray(filled), "nearest"))
            body += f"Frame {f}: " + tok * rep + "\n"
        qs = (f"There are some objects I am curious about: {self.prompt_token};\n" + body + self.question)
        prompt = ("<|begin_of_text|><|start_header_id|>user<|end_header_id|>\n\n" + tk.image_token * (rep * len(pix)) + qs +
                  "<|eot_id|><|start_header_id|>assistant<|end_header_id|>\n\n")
        enc = tk([prompt])
        dev = self.device
        ids = torch.tensor(enc["input_ids"], dtype=torch.int64)
        return dict(input_ids=ids.to(dev),
                    attention_mask=torch.ones_like(ids).to(dev).to(self.data_dtype),
                    pixel_values=torch.cat(pix, 0).to(dev).to(self.data_dtype),
                    global_mask_values=torch.cat(msk, 0).to(dev).to(self.data_dtype),
                    bboxes=[bboxes], aspect_ratios=torch.tensor([[1, 1]]).to(dev),
                    feature_replay_video=True, video_frame_tokens=frame_tokens)
