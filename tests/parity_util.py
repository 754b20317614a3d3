"""Shared by the CPU and GPU parity tests: what makes a greedy-decode comparison discriminating, and the hand-derived
known-answer cases for ``torchvision.ops.roi_align`` as the reference invokes it (modeling_gar.py:389-396).

Nothing here imports the oracle: the expected values are closed forms written out from the algorithm's published
description, so that the oracle, its C twin and the HIP kernel can each be checked against them."""
import math

import torch

F32_LOGIT_TOL = 2e-4      # max |dlogit| <= tol * max|logit| in f32 mode (north_star: fp32, stated tolerance on logits)
MARGIN_FACTOR = 10.0      # a token comparison counts only where the oracle's top-2 margin >= 10 x that tolerance


def discrimination_stats(seq_row, logits_row):
    """(distinct tokens, immediate repeats, min top-2 margin / max|logit|) of one greedy sequence."""
    t = [int(x) for x in seq_row]
    top2 = logits_row.float().topk(2, -1).values
    margin = float((top2[..., 0] - top2[..., 1]).min())
    return len(set(t)), sum(1 for a, b in zip(t, t[1:]) if a == b), margin / float(logits_row.abs().max())


def assert_discriminating(seq, logits, min_distinct_frac=0.8):
    """The oracle's greedy sequences must be able to expose a wrong decode step: no fixed point (a token followed by
    itself), mostly distinct tokens, and every step's top-2 margin well above the logit tolerance."""
    for b in range(seq.shape[0]):
        distinct, repeats, rel = discrimination_stats(seq[b].tolist(), logits[b])
        n = seq.shape[1]
        assert repeats == 0, f"row {b}: greedy sequence has an immediate repeat: {seq[b].tolist()}"
        assert distinct >= math.ceil(min_distinct_frac * n), f"row {b}: only {distinct} distinct of {n}: {seq[b].tolist()}"
        assert rel >= MARGIN_FACTOR * F32_LOGIT_TOL, f"row {b}: top-2 margin {rel:.2e} x max|logit| is within 10x the logit tolerance"


# ---- RoI-align known answers ---------------------------------------------------------------------------------------
def _interp1(f, n, x):
    """1-D rule of torchvision's bilinear_interpolate along one axis of length n: None if the sample is out of range
    (x < -1 or x > n), else clamp to >= 0, and from the last cell on return f[n-1]."""
    if x < -1.0 or x > n:
        return None
    x = max(x, 0.0)
    lo = int(x)
    if lo >= n - 1:
        return f(n - 1)
    t = x - lo
    return (1.0 - t) * f(lo) + t * f(lo + 1)


def _expected(fy, fx, H, W, roi, ss, P, aligned=True, sampling=2):
    """out[ph][pw] for the additive-separable map F[y][x] = fy(y) + fx(x): bilinear interpolation is then the sum of the
    two 1-D interpolations, a sample with either coordinate out of range contributes 0, a bin is the mean of its
    sampling x sampling samples at ((i + .5) / sampling) of the bin."""
    x1, y1, x2, y2 = roi
    off = 0.5 if aligned else 0.0
    sw, sh, ew, eh = x1 * ss - off, y1 * ss - off, x2 * ss - off, y2 * ss - off
    rw, rh = ew - sw, eh - sh
    if not aligned:
        rw, rh = max(rw, 1.0), max(rh, 1.0)
    bw, bh = rw / P, rh / P
    out = [[0.0] * P for _ in range(P)]
    for ph in range(P):
        for pw in range(P):
            acc = 0.0
            for iy in range(sampling):
                y = sh + ph * bh + (iy + 0.5) * bh / sampling
                vy = _interp1(fy, H, y)
                for ix in range(sampling):
                    x = sw + pw * bw + (ix + 0.5) * bw / sampling
                    vx = _interp1(fx, W, x)
                    if vy is not None and vx is not None:
                        acc += vy + vx
            out[ph][pw] = acc / (sampling * sampling)
    return out


def roi_kat_cases(P=16):
    """[(name, ncw, nch, channels, roi, spatial_scale, literal)] with channels = [(fy, fx)] of additive-separable maps
    over the merged (nch*P) x (ncw*P) feature map and ``literal`` = {(channel, ph, pw): value} spot values worked out
    by hand (checked against the closed form too)."""
    zero = lambda v: 0.0     # noqa: E731
    ident = lambda v: float(v)   # noqa: E731
    cases = []
    # A  aligned=True half-pixel shift: F = x, roi x in [4, 20] -> start 3.5, bin 1.0, samples 3.75 + pw, 4.25 + pw:
    #    out = 4 + pw exactly (aligned=False would give 4.5 + pw)
    cases.append(("aligned_half_pixel", 2, 2, [(zero, ident)], (4.0, 4.0, 20.0, 20.0), 1.0,
                  {(0, 0, 0): 4.0, (0, 5, 7): 11.0, (0, 15, 15): 19.0}))
    # B  sampling_ratio=2 positions at a non-integer bin size: F = x^2 on the grid (piecewise-linear between integers),
    #    roi x in [2, 14] -> start 1.5, bin 0.75; pw = 0: samples 1.6875, 2.0625 -> 1 + .6875*3 = 3.0625 and
    #    4 + .0625*5 = 4.3125 -> 3.6875; a single centre sample (1.875 -> 3.625) or samples at the bin edges differ
    cases.append(("sample_positions_x2", 2, 2, [(zero, lambda v: float(v * v))], (2.0, 2.0, 14.0, 14.0), 1.0,
                  {(0, 3, 0): 3.6875}))
    # C  y <= 0 clamp and y < -1 -> 0: F = y + 10, roi y in [-1.5, 14.5] -> start -2, bin 1: row 0 samples -1.75, -1.25
    #    are out of range (0), row 1 samples -0.75, -0.25 clamp to y = 0 (10), row 2 samples .25, .75 (10.5)
    cases.append(("top_clamp", 2, 2, [(lambda v: float(v) + 10.0, zero)], (4.0, -1.5, 20.0, 14.5), 1.0,
                  {(0, 0, 3): 0.0, (0, 1, 3): 10.0, (0, 2, 3): 10.5, (0, 9, 0): 17.5}))
    # D  last-cell branch (yl >= H-1) and y > H -> 0: H = 32, F = y, roi y in [17.5, 33.5] -> start 17, bin 1:
    #    row 13 samples 30.25, 30.75 (30.5); row 14 samples 31.25, 31.75 are in the last cell (31); row 15 samples
    #    32.25, 32.75 are beyond H (0)
    cases.append(("bottom_edge", 2, 2, [(ident, zero)], (4.0, 17.5, 20.0, 33.5), 1.0,
                  {(0, 13, 0): 30.5, (0, 14, 0): 31.0, (0, 15, 0): 0.0}))
    # E  a box straddling two tiles of the tile layout (ncw = 2, nch = 1): channel 0 is 1 on the left tile and 3 on
    #    the right tile, channel 1 = x, channel 2 = y. roi x in [12.5, 20.5] -> start 12, bin .5: pw = 6 samples 15.125,
    #    15.375 -> 1.25, 1.75 (1.5); pw = 7 -> 2.5; pw <= 5 -> 1; pw >= 8 -> 3
    step = lambda v: 1.0 if v <= 15 else 3.0    # noqa: E731
    cases.append(("two_tiles", 2, 1, [(zero, step), (zero, ident), (ident, zero)], (12.5, 2.5, 20.5, 10.5), 1.0,
                  {(0, 0, 5): 1.0, (0, 0, 6): 1.5, (0, 0, 7): 2.5, (0, 0, 8): 3.0, (1, 4, 6): 15.25, (2, 4, 6): 4.25}))
    # F  the reference's double scaling: roi pre-multiplied by 28, spatial_scale 1/28 (case A's geometry)
    cases.append(("double_scaled", 2, 2, [(zero, ident)], (112.0, 112.0, 560.0, 560.0), 1.0 / 28.0,
                  {(0, 0, 0): 4.0, (0, 5, 7): 11.0}))
    return cases


def kat_feature_map(case, P=16):
    """(fmap [C, H, W] float32, expected [C, P, P] float64 tensor) of one case."""
    name, ncw, nch, chans, roi, ss, literal = case
    H, W = nch * P, ncw * P
    fmap = torch.zeros(len(chans), H, W, dtype=torch.float32)
    exp = torch.zeros(len(chans), P, P, dtype=torch.float64)
    for c, (fy, fx) in enumerate(chans):
        for y in range(H):
            for x in range(W):
                fmap[c, y, x] = fy(y) + fx(x)
        exp[c] = torch.tensor(_expected(fy, fx, H, W, roi, ss, P), dtype=torch.float64)
    for (c, ph, pw), v in literal.items():          # the hand-worked spot values agree with the closed form
        assert abs(float(exp[c, ph, pw]) - v) < 1e-4, (name, c, ph, pw, float(exp[c, ph, pw]), v)
    return fmap, exp


def tiles_from_map(fmap, ncw, nch, P=16, thumbnail_fill=1.0e9):
    """[C, nch*P, ncw*P] -> the tile layout the model holds, [1 + ncw*nch, P*P, C]: tile 0 is the thumbnail (filled with
    a huge value: it must never be read), map cell (y, x) = token (y % P) * P + x % P of tile 1 + (y // P) * ncw + x // P
    (modeling_gar.py:248-260, pinned by tests/golden/ref_helpers.json)."""
    C = fmap.shape[0]
    t = torch.full((1 + ncw * nch, P * P, C), thumbnail_fill, dtype=torch.float32)
    for y in range(nch * P):
        for x in range(ncw * P):
            t[1 + (y // P) * ncw + x // P, (y % P) * P + x % P] = fmap[:, y, x]
    return t


# ---- timm RotaryEmbeddingCat, written out per position -------------------------------------------------------------
def rope2d_table_loops(head_dim, grid, temperature=10000.0, offset=1.0, indexing="xy"):
    """sin / cos [grid*grid, head_dim] from the definition: bands_k = T^(-k / (hd/4)); patch (row i,
    col j); with 'xy' the first hd/2 entries rotate by the x coordinate (j + offset), the second hd/2 by y (i + offset)
    — 'ij' the other way round —, every angle repeated for the two members of an interleaved pair."""
    nb = head_dim // 4
    bands = [temperature ** (-(k / nb)) for k in range(nb)]
    sin = torch.zeros(grid * grid, head_dim, dtype=torch.float64)
    cos = torch.zeros(grid * grid, head_dim, dtype=torch.float64)
    for i in range(grid):
        for j in range(grid):
            first, second = (j + offset, i + offset) if indexing == "xy" else (i + offset, j + offset)
            for k in range(nb):
                for rep in range(2):
                    sin[i * grid + j, 2 * k + rep] = math.sin(first * bands[k])
                    cos[i * grid + j, 2 * k + rep] = math.cos(first * bands[k])
                    sin[i * grid + j, 2 * nb + 2 * k + rep] = math.sin(second * bands[k])
                    cos[i * grid + j, 2 * nb + 2 * k + rep] = math.cos(second * bands[k])
    return sin, cos
