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


def _poly_runs(xy, h: int, w: int):
    """Run lengths (column-major, starting with background) of one polygon, following the published COCO mask API
    rasteriser (`rleFrPoly`, pycocotools — not available here, so this restatement is PARITY UNPINNED): the polygon is
    up-sampled x5, its boundary walked with one point per unit step along the longer axis, boundary points where the
    down-sampled x lands on a pixel centre column become (column, first-row) crossings, and the sorted crossings are the
    run boundaries."""
    scale = 5.0
    k = len(xy) // 2
    x = [int(scale * xy[2 * j] + 0.5) for j in range(k)]
    y = [int(scale * xy[2 * j + 1] + 0.5) for j in range(k)]
    x.append(x[0])
    y.append(y[0])
    u, v = [], []
    for j in range(k):
        xs, xe, ys, ye = x[j], x[j + 1], y[j], y[j + 1]
        dx, dy = abs(xe - xs), abs(ys - ye)
        flip = (dx >= dy and xs > xe) or (dx < dy and ys > ye)
        if flip:
            xs, xe, ys, ye = xe, xs, ye, ys
        s = (ye - ys) / dx if dx >= dy and dx else ((xe - xs) / dy if dy else 0.0)
        if dx >= dy:
            for d in range(dx + 1):
                t = dx - d if flip else d
                u.append(t + xs)
                v.append(int(ys + s * t + 0.5))
        else:
            for d in range(dy + 1):
                t = dy - d if flip else d
                v.append(t + ys)
                u.append(int(xs + s * t + 0.5))
    pts = []
    for j in range(1, len(u)):
        if u[j] != u[j - 1]:
            xd = float(u[j] if u[j] < u[j - 1] else u[j] - 1)
            xd = (xd + 0.5) / scale - 0.5
            if np.floor(xd) != xd or xd < 0 or xd > w - 1:
                continue
            yd = float(v[j] if v[j] < v[j - 1] else v[j - 1])
            yd = (yd + 0.5) / scale - 0.5
            yd = min(max(yd, 0.0), float(h))
            pts.append(int(xd) * h + int(np.ceil(yd)))
    pts.append(h * w)
    pts.sort()
    runs, p = [], 0
    for a in pts:
        runs.append(a - p)
        p = a
    merged, j = [runs[0]], 1        # fold zero-length runs into their neighbours
    while j < len(runs):
        if runs[j] > 0:
            merged.append(runs[j])
            j += 1
        else:
            j += 1
            if j < len(runs):
                merged[-1] += runs[j]
                j += 1
    return merged


def from_polygons(polys, h: int, w: int) -> np.ndarray:
    """List of polygons ([x0, y0, x1, y1, ...] each) -> uint8 [h, w] mask of their union, i.e. what the reference's
    `annToMask` (evaluation/Ferret-Bench/inference.py:67-71: frPyObjects -> merge -> decode) returns."""
    out = np.zeros((h, w), dtype=np.uint8)
    for xy in polys:
        out |= decode({"size": [h, w], "counts": _poly_runs(list(xy), h, w)})
    return out
