"""Seeded synthetic inputs of the benchmark workload: uint8 1024x1024 images, random
rectangle-union-ellipse masks, multi-region disjoint masks. No dataset or checkpoint is reachable here."""
from __future__ import annotations

import numpy as np
from PIL import Image


def synthetic_image(i: int, width: int = 1024, height: int = 1024) -> Image.Image:
    rng = np.random.default_rng(1000 + i)
    return Image.fromarray(rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8), "RGB")


def synthetic_mask(i: int, width: int = 1024, height: int = 1024, lo: int = 64, hi: int = 512) -> np.ndarray:
    """Rectangle with uniform random corner/size in [lo,hi]^2 px unioned with a random ellipse; non-empty."""
    rng = np.random.default_rng(2000 + i)
    lo = min(lo, max(1, width // 8), max(1, height // 8))
    hi = max(lo + 1, min(hi, width // 2, height // 2))
    m = np.zeros((height, width), dtype=bool)
    w, h = int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))
    x0, y0 = int(rng.integers(0, width - w + 1)), int(rng.integers(0, height - h + 1))
    m[y0:y0 + h, x0:x0 + w] = True
    cx, cy = rng.uniform(0, width), rng.uniform(0, height)
    ax, ay = rng.uniform(lo / 2, hi / 2), rng.uniform(lo / 2, hi / 2)
    yy, xx = np.mgrid[0:height, 0:width]
    m |= ((xx - cx) / ax) ** 2 + ((yy - cy) / ay) ** 2 <= 1.0
    return m


def synthetic_disjoint_masks(i: int, n: int = 4, width: int = 1024, height: int = 1024):
    """n disjoint rectangles (one per quadrant-like cell), for the multi-region relationship prompt."""
    rng = np.random.default_rng(3000 + i)
    cols = int(np.ceil(np.sqrt(n)))
    rows = int(np.ceil(n / cols))
    cw, ch = width // cols, height // rows
    out = []
    for k in range(n):
        r, c = divmod(k, cols)
        w, h = int(rng.integers(cw // 4, cw // 2 + 1)), int(rng.integers(ch // 4, ch // 2 + 1))
        x0 = c * cw + int(rng.integers(0, cw - w + 1))
        y0 = r * ch + int(rng.integers(0, ch - h + 1))
        m = np.zeros((height, width), dtype=bool)
        m[y0:y0 + h, x0:x0 + w] = True
        out.append(m)
    return out


RELATIONSHIP_QUESTION = ("Question: What is the relationship between <Prompt0>, <Prompt1>, <Prompt2>, and <Prompt3>?\n"
                         "Options:\nA. <Prompt0> is holding <Prompt1>\nB. <Prompt2> is next to <Prompt3>\n"
                         "C. They are unrelated\nD. <Prompt1> is on <Prompt2>\n"
                         "Answer with the correct option's letter directly.")
