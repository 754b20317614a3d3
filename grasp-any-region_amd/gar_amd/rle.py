"""COCO run-length masks: decode through ``gar_rle_decode`` (libgar_hip.so, host C++); a small pure-Python ENCODER for
tests and synthetic annotation files. Format restated in csrc/core.hip."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import hip


def decode(rle: dict) -> np.ndarray:
    """{"size": [h, w], "counts": str | bytes | list[int]} -> uint8 [h, w] (0 / 1), like pycocotools.mask.decode."""
    h, w = int(rle["size"][0]), int(rle["size"][1])
    counts = rle["counts"]
    if isinstance(counts, (list, tuple)):                 # uncompressed RLE: plain run lengths, column-major
        flat = np.zeros(h * w, dtype=np.uint8)
        pos, val = 0, 0
        for c in counts:
            if val:
                flat[pos:pos + int(c)] = 1
            pos += int(c)
            val ^= 1
        if pos != h * w:
            raise hip.GarError(f"rle: runs cover {pos} of {h * w} pixels")
        return np.ascontiguousarray(flat.reshape(w, h).T)
    if isinstance(counts, str):
        counts = counts.encode("ascii")
    out = np.empty((h, w), dtype=np.uint8)
    n = hip.load_library().gar_rle_decode(counts, len(counts), h, w, out.ctypes.data_as(C.c_void_p))
    if n < 0:
        raise hip.GarError("gar_rle_decode: " + hip.load_library().gar_last_error().decode())
    return out


def encode(mask: np.ndarray) -> dict:
    """uint8/bool [h, w] -> compressed COCO RLE dict (test / synthetic-data helper)."""
    m = np.asarray(mask).astype(bool)
    h, w = m.shape
    flat = m.T.reshape(-1)                                # column-major pixel order
    change = np.flatnonzero(flat[1:] != flat[:-1]) + 1
    bounds = np.concatenate(([0], change, [flat.size]))
    runs = list(np.diff(bounds))
    if flat.size and flat[0]:
        runs = [0] + runs                                 # runs start with background
    chars = []
    for i, c in enumerate(runs):
        x = int(c) - (int(runs[i - 2]) if i > 2 else 0)
        more = True
        while more:
            g = x & 0x1f
            x >>= 5
            more = not ((x == 0 and not (g & 0x10)) or (x == -1 and (g & 0x10)))
            if more:
                g |= 0x20
            chars.append(chr(g + 48))
    return {"size": [h, w], "counts": "".join(chars)}
