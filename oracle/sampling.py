"""CPU ORACLE — TEST INFRASTRUCTURE ONLY (see gar_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import anything under oracle/).

``do_sample=True`` of the generation the reference forwards to (``/root/reference/projects/grasp_any_region/hf_models/
modeling_gar.py:418-426`` passes the caller's ``GenerationConfig`` to ``GenerationMixin.generate``): the logits warpers of
transformers==4.56.2 ``generation/logits_process.py`` in the order ``GenerationMixin._get_logits_processor`` appends them
(``TemperatureLogitsWarper`` -> ``TopKLogitsWarper`` -> ``TopPLogitsWarper``) and one draw from ``softmax`` of what is left
(``GenerationMixin._sample``: ``torch.multinomial(probs, 1)``).

PINNED (kept set): ``warp`` against the installed transformers' three warper classes, ``tests/test_oracle_goldens.py``.
UNPINNED by construction (the draw): HF draws with torch's global generator; a device loop draws with a counter-based generator so
that a captured hipGraph replays it. What is pinned instead: ``philox4x32_10`` against the Random123 known-answer vectors, and the
draw's definition (inverse CDF in vocabulary order) against the empirical distribution of ``torch.multinomial``.
"""
from __future__ import annotations

import numpy as np

PHILOX_M0, PHILOX_M1 = 0xD2511F53, 0xCD9E8D57
PHILOX_W0, PHILOX_W1 = 0x9E3779B9, 0xBB67AE85


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3", SC'11). counter: 4 words, key: 2 words."""
    c = [int(x) & 0xFFFFFFFF for x in counter]
    k = [int(x) & 0xFFFFFFFF for x in key]
    for _ in range(10):
        p0, p1 = PHILOX_M0 * c[0], PHILOX_M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k[0]) & 0xFFFFFFFF, p1 & 0xFFFFFFFF, ((p0 >> 32) ^ c[3] ^ k[1]) & 0xFFFFFFFF, p0 & 0xFFFFFFFF]
        k = [(k[0] + PHILOX_W0) & 0xFFFFFFFF, (k[1] + PHILOX_W1) & 0xFFFFFFFF]
    return c


def uniform(seed: int, row: int, step: int) -> float:
    """u in [0, 1): the top 24 bits of the first Philox word of counter (step, row, 0, 0) under key (seed lo, seed hi)."""
    r = philox4x32_10((step, row, 0, 0), (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))[0]
    return (r >> 8) / 16777216.0


def warp(logits: np.ndarray, temperature: float = 1.0, top_k: int = 0, top_p: float = 1.0):
    """(z, keep): z = logits / temperature (float32, as ``scores / self.temperature``); keep = the tokens the top-k and top-p
    warpers leave (the others are set to -inf by HF).

    TopKLogitsWarper: ``scores < topk(scores, k)[0][..., -1, None]`` is removed — ties with the k-th value stay.
    TopPLogitsWarper: ascending sort, ``cumsum(softmax) <= 1 - top_p`` is removed, the last (largest) always stays — a token stays
    iff the probability of the tokens sorted behind it (the larger ones) is < top_p. Equal logits are taken together here (the
    sort would cut through them in an unspecified order)."""
    z = (logits.astype(np.float32) / np.float32(temperature)).astype(np.float32)
    V = z.shape[-1]
    keep = np.ones(V, dtype=bool)
    if 0 < top_k < V:
        kth = np.sort(z)[V - top_k]
        keep &= z >= kth
    if top_p < 1.0:
        m = z[keep].max()
        e = np.where(keep, np.exp((z - m).astype(np.float64)), 0.0)
        Z1 = e.sum()
        order = np.argsort(-z, kind="stable")
        es = e[order]
        zs = z[order]
        above = np.concatenate([[0.0], np.cumsum(es)[:-1]])          # mass sorted in front of each token (ties included so far)
        # mass of the STRICTLY larger logits: the value of `above` at the first token of each run of equal logits
        first = np.r_[True, zs[1:] != zs[:-1]]
        above_strict = np.maximum.accumulate(np.where(first, above, -1.0))
        stay_sorted = above_strict < top_p * Z1
        stay = np.zeros(V, dtype=bool)
        stay[order] = stay_sorted
        keep &= stay
    return z, keep


def draw(z: np.ndarray, keep: np.ndarray, u: float):
    """(token, margin): the first index in vocabulary order whose running sum of kept exp(z - max) exceeds u x their total, and how
    far (relative to the total) the target lies from the nearest boundary of that token's interval — a device result computed in
    float32 may legitimately land on the neighbouring kept token when the margin is within rounding."""
    m = z[keep].max()
    e = np.where(keep, np.exp((z - m).astype(np.float64)), 0.0)
    c = np.cumsum(e)
    target = u * c[-1]
    tok = int(np.searchsorted(c, target, side="right"))
    tok = min(tok, len(z) - 1)
    while not keep[tok]:           # searchsorted can stop on a removed token whose cumulative value equals the target
        tok += 1
    lo = c[tok] - e[tok]
    margin = min(target - lo, c[tok] - target) / c[-1]
    return tok, float(margin)


def sample(logits: np.ndarray, temperature: float, top_k: int, top_p: float, seed: int, row: int, step: int):
    z, keep = warp(logits, temperature, top_k, top_p)
    return draw(z, keep, uniform(seed, row, step)) + (keep,)
