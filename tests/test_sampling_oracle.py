"""CPU: the sampling oracle (oracle/sampling.py) against what it restates — transformers' logits warpers (kept set), the
Random123 known-answer vectors (Philox4x32-10) and torch.multinomial (the draw's distribution)."""
import numpy as np
import pytest
import torch

from oracle import sampling as S


def test_philox_known_answers():
    # Random123 kat_vectors, philox4x32 10 rounds
    assert S.philox4x32_10((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert S.philox4x32_10((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert S.philox4x32_10((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]
    u = [S.uniform(1234, r, s) for r in range(4) for s in range(64)]
    assert all(0.0 <= x < 1.0 for x in u) and len(set(u)) == len(u)
    assert 0.35 < float(np.mean(u)) < 0.65


@pytest.mark.parametrize("temperature,top_k,top_p", [(1.0, 0, 1.0), (0.7, 50, 1.0), (1.3, 0, 0.9), (0.6, 40, 0.8), (1.0, 5, 0.5),
                                                      (2.0, 1000, 0.99), (1.0, 1, 1.0)])
def test_kept_set_equals_transformers_warpers(temperature, top_k, top_p):
    lp = pytest.importorskip("transformers.generation.logits_process")
    g = torch.Generator().manual_seed(5)
    for V in (97, 4096):
        logits = (torch.randn(3, V, generator=g) * 3.0).float()
        scores = logits.clone()
        ids = torch.zeros(3, 1, dtype=torch.long)
        if temperature != 1.0:
            scores = lp.TemperatureLogitsWarper(temperature)(ids, scores)
        if top_k:
            scores = lp.TopKLogitsWarper(top_k)(ids, scores)
        if top_p < 1.0:
            scores = lp.TopPLogitsWarper(top_p)(ids, scores)
        for b in range(3):
            z, keep = S.warp(logits[b].numpy(), temperature, top_k, top_p)
            hf_keep = torch.isfinite(scores[b]).numpy()
            assert np.array_equal(keep, hf_keep), (V, b, int(keep.sum()), int(hf_keep.sum()))
            np.testing.assert_allclose(z[keep], scores[b].numpy()[hf_keep], rtol=0, atol=0)


def test_ties_stay_together():
    logits = np.array([0.0, 2.0, 2.0, 2.0, -1.0, 1.0], dtype=np.float32)
    _, keep = S.warp(logits, 1.0, top_k=2, top_p=1.0)
    assert keep.tolist() == [False, True, True, True, False, False]        # HF: scores < k-th value removed, ties at it stay
    _, keep = S.warp(logits, 1.0, top_k=0, top_p=0.3)
    assert keep.tolist() == [False, True, True, True, False, False]


def test_draw_is_the_inverse_cdf_and_matches_multinomial():
    logits = np.array([1.0, -0.5, 0.25, 3.0, -2.0, 0.0, 2.0, 1.5], dtype=np.float32)
    z, keep = S.warp(logits, 0.8, top_k=6, top_p=0.95)
    p = np.where(keep, np.exp(z.astype(np.float64) - z[keep].max()), 0.0)
    p /= p.sum()
    n = 20000
    toks = np.array([S.draw(z, keep, S.uniform(77, 0, s))[0] for s in range(n)])
    assert keep[toks].all()
    freq = np.bincount(toks, minlength=len(logits)) / n
    ref = np.bincount(torch.multinomial(torch.from_numpy(p), n, replacement=True, generator=torch.Generator().manual_seed(0)).numpy(),
                      minlength=len(logits)) / n
    assert np.abs(freq - p).max() < 0.012, (freq, p)
    assert np.abs(ref - p).max() < 0.012
    # u = 0 takes the first kept token, u -> 1 the last one
    assert S.draw(z, keep, 0.0)[0] == int(np.flatnonzero(keep)[0])
    assert S.draw(z, keep, 1.0 - 2.0 ** -24)[0] in np.flatnonzero(keep)[-2:]
