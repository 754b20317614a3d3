"""Device-side image preprocessing.

``GpuImageProcessor`` is a drop-in for ``processing.GARImageProcessor`` (same call signature and return value —
reference contract: image_processing_perception_lm_fast.py:268-372) that uploads the raw uint8 RGB image once and
produces the thumbnail + tiles directly on the GPU in the model dtype. The host keeps what is host work in the
reference too: canvas selection and the (tiny, cached) resampling tables.

Tables. The antialiased-bicubic taps are the ones torch's fp32 ``upsample_bicubic2d_aa`` uses: the band
(first tap, tap count) follows its published formula and the normalised weights are read back from the operator itself
by resizing an identity matrix once per (in, out) size pair — an impulse through a linear resampler returns its weight,
so the device result is bit-identical to the host processor (tests/test_gpu_preprocess.py), which a re-derivation of
the weights in Python is not (few-ulp differences flip ~0.005 % of the rounded uint8 pixels). NEAREST index tables are
obtained the same way from ``mode="nearest"``.
"""
from __future__ import annotations

import functools
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F
from PIL import Image

from . import hip, ops
from .processing import select_canvas


@functools.lru_cache(maxsize=64)
def bicubic_aa_taps(in_size: int, out_size: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """(first[out], count[out], weights[out, kmax]) of torch's antialiased bicubic resampling in -> out (fp32)."""
    f32 = np.float32
    scale = f32(f32(in_size) / f32(out_size))
    support = f32(f32(2.0) * scale) if scale >= 1.0 else f32(2.0)
    kmax = int(np.ceil(support)) * 2 + 1
    first = np.zeros(out_size, np.int32)
    count = np.zeros(out_size, np.int32)
    for i in range(out_size):
        center = f32(float(scale) * (i + 0.5))
        lo = max(int(float(center) - float(support) + 0.5), 0)
        n = min(int(float(center) + float(support) + 0.5), in_size) - lo
        first[i], count[i] = lo, min(max(n, 0), kmax)
    eye = torch.eye(in_size, dtype=torch.float32).reshape(1, 1, in_size, in_size)
    full = F.interpolate(eye, size=(in_size, out_size), mode="bicubic", align_corners=False, antialias=True)[0, 0].numpy()
    w = np.zeros((out_size, kmax), f32)
    inside = np.zeros_like(full, dtype=bool)
    for i in range(out_size):
        w[i, :count[i]] = full[first[i]:first[i] + count[i], i]
        inside[first[i]:first[i] + count[i], i] = True
    if np.any(full[~inside] != 0):          # the band formula must cover every non-zero tap of the operator
        raise hip.GarError(f"bicubic tap band mismatch for {in_size}->{out_size}")
    return first, count, w


@functools.lru_cache(maxsize=64)
def nearest_index(in_size: int, out_size: int) -> np.ndarray:
    src = torch.arange(in_size, dtype=torch.float32).reshape(1, 1, 1, in_size)
    return F.interpolate(src, size=(1, out_size), mode="nearest")[0, 0, 0].to(torch.int32).numpy()


class GpuImageProcessor:
    """thumb + tile preprocessing on the GPU; mean = std = 0.5, RGB conversion on the host (PIL)."""

    def __init__(self, tile_size: int = 448, max_num_tiles: int = 16, resample: str = "bicubic", device="cuda:0",
                 dtype: torch.dtype = torch.bfloat16):
        self.tile_size = tile_size
        self.max_num_tiles = max_num_tiles
        self.resample = resample
        self.image_mean = 0.5
        self.image_std = 0.5
        self.device = torch.device(device)
        self.dtype = dtype
        hip.require_device(self.device.index or 0)
        self._tables = {}
        self._tmp: Optional[torch.Tensor] = None

    def _taps(self, in_size, out_size):
        key = ("b", in_size, out_size)
        if key not in self._tables:
            self._tables[key] = tuple(torch.from_numpy(a).to(self.device) for a in bicubic_aa_taps(in_size, out_size))
        return self._tables[key]

    def _index(self, in_size, out_size):
        key = ("n", in_size, out_size)
        if key not in self._tables:
            self._tables[key] = torch.from_numpy(nearest_index(in_size, out_size)).to(self.device)
        return self._tables[key]

    def _resize_into(self, src, out, n_w, n_h, tile0, resample):
        ts = self.tile_size
        H, W, _ = src.shape
        Wout, Hout = n_w * ts, n_h * ts
        if resample == "nearest":
            ops.resize_nearest_tiles(src, out, ts, n_w, tile0, self._index(W, Wout), self._index(H, Hout),
                                     self.image_mean, self.image_std)
            return
        need = 3 * H * Wout
        if self._tmp is None or self._tmp.numel() < need:
            self._tmp = torch.empty(need, dtype=torch.float32, device=self.device)
        ops.resize_bicubic_tiles(src, self._tmp, out, ts, n_w, tile0, self._taps(W, Wout), self._taps(H, Hout),
                                 self.image_mean, self.image_std)

    def _upload(self, image: Image.Image) -> torch.Tensor:
        rgb = np.array(image.convert("RGB"), dtype=np.uint8)                        # writable, contiguous copy
        return torch.from_numpy(rgb).to(self.device, non_blocking=True)             # [H, W, 3] uint8

    def single_tile(self, image: Image.Image, resample: Optional[str] = None) -> torch.Tensor:
        with torch.cuda.device(self.device):            # kernels go to the current HIP device / its current stream
            return self._single_tile(image, resample)

    def _single_tile(self, image, resample):
        out = torch.empty(1, 3, self.tile_size, self.tile_size, dtype=self.dtype, device=self.device)
        self._resize_into(self._upload(image), out, 1, 1, 0, resample or self.resample)
        return out

    def __call__(self, image: Image.Image, resample: Optional[str] = None):
        with torch.cuda.device(self.device):
            return self._call(image, resample)

    def _call(self, image, resample):
        resample = resample or self.resample
        src = self._upload(image)
        h, w = src.shape[:2]
        ts = self.tile_size
        n_w, n_h = select_canvas(w, h, ts, self.max_num_tiles)
        out = torch.empty(1 + n_w * n_h, 3, ts, ts, dtype=self.dtype, device=self.device)
        self._resize_into(src, out, 1, 1, 0, resample)                 # thumbnail = tile 0
        self._resize_into(src, out, n_w, n_h, 1, resample)             # canvas tiles, tile = h_idx * n_w + w_idx
        return out.unsqueeze(0), [n_w, n_h]                            # [1, T+1, 3, ts, ts]
