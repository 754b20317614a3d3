"""GPU (-m gpu): do_sample = True — gar_sample (temperature / top-k / top-p warpers + one Philox draw per row and step) against
oracle/sampling.py on the same logits, and GARModel.generate with a sampling GenerationConfig against the oracle's draw from the
logits the run itself returns. Reference behaviour: modeling_gar.py:418-426 forwards the caller's GenerationConfig to HF's generate."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

MARGIN = 2e-5        # a float32 running sum of <= 128 k terms against the oracle's float64: the draw may sit this close to a boundary


def _run_kernel(logits, temperature, top_k, top_p, seed, step, eos=None):
    from gar_amd import ops
    B, V = logits.shape
    dev = logits.device
    out = torch.full((B, 8), -1, dtype=torch.int64, device=dev)
    cur = torch.zeros(B, dtype=torch.int64, device=dev)
    stepd = torch.tensor([step], dtype=torch.int32, device=dev)
    params = torch.tensor([temperature, top_p, float(top_k), 0.0], dtype=torch.float32, device=dev)
    seedd = torch.tensor([seed], dtype=torch.int64, device=dev)
    kw = {}
    if eos is not None:
        kw = dict(eos_ids=torch.tensor(eos, dtype=torch.int64, device=dev), finished=torch.full((B,), -1, dtype=torch.int32, device=dev),
                  done_count=torch.zeros(1, dtype=torch.int32, device=dev))
    ops.sample(logits, V, out, out.stride(0), stepd, cur, params, seedd, **kw)
    torch.cuda.synchronize()
    assert torch.equal(out[:, step], cur)
    return cur.cpu().tolist(), kw


def _check_against_oracle(logits, toks, temperature, top_k, top_p, seed, step):
    from oracle import sampling as S
    lg = logits.float().cpu().numpy()
    close = 0
    for b, tok in enumerate(toks):
        exp, margin, keep = S.sample(lg[b], temperature, top_k, top_p, seed, b, step)
        assert keep[tok], (b, tok, "token outside the kept set")
        if tok != exp:
            kept = np.flatnonzero(keep)
            i, j = int(np.searchsorted(kept, tok)), int(np.searchsorted(kept, exp))
            assert margin < MARGIN and abs(i - j) == 1, (b, tok, exp, margin)
            close += 1
    return close


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("V", [128256, 1000, 37])
def test_sample_kernel_equals_oracle(dtype, V):
    from gar_amd import hip
    hip.require_device(0)
    g = torch.Generator().manual_seed(V)
    B = 8
    logits = (torch.randn(B, V, generator=g) * 2.5).to(dtype).to("cuda:0")
    near = total = 0
    for temperature, top_k, top_p in [(1.0, 0, 1.0), (0.7, 50, 1.0), (1.3, 0, 0.9), (0.6, 40, 0.8), (1.0, 5, 0.5), (2.0, 20000, 0.999)]:
        for seed, step in [(1, 0), (123456789012345, 3), (2 ** 61 + 7, 5)]:
            toks, _ = _run_kernel(logits, temperature, min(top_k, V), top_p, seed, step)
            near += _check_against_oracle(logits, toks, temperature, min(top_k, V), top_p, seed, step)
            total += B
    assert near <= max(1, total // 100), (near, total)


def test_sample_kernel_properties():
    from gar_amd import hip
    hip.require_device(0)
    g = torch.Generator().manual_seed(3)
    B, V = 16, 4096
    logits = (torch.randn(B, V, generator=g) * 3.0).to("cuda:0")
    # top_k = 1 is greedy whatever the draw; so is a vanishing top_p
    am = logits.argmax(1).cpu().tolist()
    assert _run_kernel(logits, 0.9, 1, 1.0, 5, 0)[0] == am
    assert _run_kernel(logits, 1.0, 0, 1e-6, 5, 1)[0] == am
    # the same (seed, step) repeats; another seed or step moves some rows
    a = _run_kernel(logits, 1.0, 0, 1.0, 42, 2)[0]
    assert _run_kernel(logits, 1.0, 0, 1.0, 42, 2)[0] == a
    assert _run_kernel(logits, 1.0, 0, 1.0, 43, 2)[0] != a and _run_kernel(logits, 1.0, 0, 1.0, 42, 3)[0] != a
    # suppressed tokens (-inf) are never drawn; a row with ONE finite logit returns it
    lg = torch.full((B, V), float("-inf"), device="cuda:0")
    lg[torch.arange(B), torch.arange(B) * 7 + 1] = 0.5
    assert _run_kernel(lg, 1.0, 50, 0.9, 9, 0)[0] == [b * 7 + 1 for b in range(B)]
    # the eos latch is gar_argmax's: rows whose draw is an eos id are latched at this step and counted
    toks, kw = _run_kernel(logits, 1.0, 0, 1.0, 42, 2, eos=[a[0], a[5], -1])
    fin = kw["finished"].cpu().tolist()
    hit = [t in (a[0], a[5]) for t in toks]
    assert toks == a and [f == 2 for f in fin] == hit and int(kw["done_count"].item()) == sum(hit)
    # the empirical distribution of the draws of one row over many steps follows softmax(z) on the kept set
    from oracle import sampling as S
    row = (torch.randn(1, 64, generator=g) * 1.5).to("cuda:0")
    z, keep = S.warp(row[0].cpu().numpy(), 0.8, 20, 0.95)
    p = np.where(keep, np.exp(z.astype(np.float64) - z[keep].max()), 0.0)
    p /= p.sum()
    n = 4000
    counts = np.zeros(64)
    big = row.expand(64, 64).contiguous()                      # 64 rows of the same logits: 64 independent draws per launch
    for s in range(n // 64):
        for t in _run_kernel(big, 0.8, 20, 0.95, 2024 + s, s % 8)[0]:
            counts[t] += 1
    assert counts[~keep].sum() == 0
    assert np.abs(counts / counts.sum() - p).max() < 0.03, (counts / counts.sum(), p)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_generate_with_sampling_config(dtype):
    """GARModel.generate(generation_config = do_sample ...): every token of every row is the oracle's draw (same seed, row, step)
    from the logits the run returns for that step; the hipGraph loop and the eager loop give the same tokens; a transformers
    GenerationConfig object and a dict are read alike; greedy calls on the same model are untouched."""
    from gar_amd import GARConfig
    from gar_amd.eval_dataset import SingleRegionCaptionDataset
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    from gar_amd.weights import synthetic_weights
    from oracle import sampling as S
    cfg = GARConfig.tiny()
    proc = GARProcessor.from_config(cfg, max_num_tiles=4)
    m = GARModel(cfg, synthetic_weights(cfg), dtype)
    smp = [SingleRegionCaptionDataset(synthetic_image(i, 200, 160), synthetic_mask(i, 200, 160), proc, data_dtype=dtype, device="cpu")[0]
           for i in range(3)]
    batch = {k: torch.cat([s[k] for s in smp], 0) if isinstance(smp[0][k], torch.Tensor) else [x for s in smp for x in s[k]]
             for k in smp[0]}
    V = cfg.mllm_config.text_config.vocab_size
    NT = 12
    gc = dict(do_sample=True, temperature=0.8, top_k=min(50, V), top_p=0.9, max_new_tokens=NT)
    greedy0 = m.generate(**batch, max_new_tokens=NT).sequences.cpu()
    out = m.generate(**batch, generation_config=gc, seed=777, return_logits=True)
    seq, lg = out.sequences.cpu(), out.logits.float().cpu().numpy()
    assert tuple(seq.shape) == (3, NT)
    near = 0
    for b in range(3):
        for j in range(NT):
            exp, margin, keep = S.sample(lg[b, j, :V], 0.8, min(50, V), 0.9, 777, b, j)
            tok = int(seq[b, j])
            assert keep[tok]
            if tok != exp:
                assert margin < MARGIN, (b, j, tok, exp, margin)
                near += 1
    assert near <= 1
    assert not torch.equal(seq, greedy0)                                         # it does sample
    eager = m.generate(**batch, generation_config=gc, seed=777, use_graph=False).sequences.cpu()
    assert torch.equal(eager, seq)
    assert not torch.equal(m.generate(**batch, generation_config=gc, seed=778).sequences.cpu(), seq)
    torch.manual_seed(11)
    a = m.generate(**batch, generation_config=gc).sequences.cpu()                # seed drawn from torch's global generator
    torch.manual_seed(11)
    assert torch.equal(m.generate(**batch, generation_config=gc).sequences.cpu(), a)
    try:
        from transformers import GenerationConfig
        obj = GenerationConfig(do_sample=True, temperature=0.8, top_k=min(50, V), top_p=0.9, max_new_tokens=NT)
        assert torch.equal(m.generate(**batch, generation_config=obj, seed=777).sequences.cpu(), seq)
    except ImportError:
        pass
    assert torch.equal(m.generate(**batch, max_new_tokens=NT).sequences.cpu(), greedy0)       # greedy graph of the same bucket intact
    from gar_amd import hip
    with pytest.raises(hip.GarError):
        m.generate(**batch, generation_config=dict(do_sample=True, typical_p=0.5))
    with pytest.raises(ValueError):
        m.generate(**batch, generation_config=dict(do_sample=True, temperature=0.0))
    # a batch whose prompt phase runs in several chunks draws what the un-chunked batch draws: the first tokens are counted by the
    # GLOBAL batch row (gar_sample's row_offset, ABI 15 — ADVICE r5: rows i and b0 + i used to share their step-0 draw)
    m1 = GARModel(cfg, synthetic_weights(cfg), dtype, prefill_chunk=1)
    chunked = m1.generate(**batch, generation_config=gc, seed=777).sequences.cpu()
    assert torch.equal(chunked, seq)
    same = {k: (torch.cat([smp[0][k]] * 3, 0) if isinstance(smp[0][k], torch.Tensor) else [x for _ in range(3) for x in smp[0][k]])
            for k in smp[0]}
    rows = m1.generate(**same, generation_config=gc, seed=5).sequences.cpu()        # identical prompts: the rows must still differ
    assert len({tuple(r.tolist()) for r in rows}) == 3
    # options passed as keywords are laid over the config (HF generate semantics), never swallowed
    assert torch.equal(m.generate(**batch, do_sample=True, temperature=0.8, top_k=min(50, V), top_p=0.9, max_new_tokens=NT,
                                  seed=777).sequences.cpu(), seq)
    with pytest.raises(hip.GarError, match="num_beams"):
        m.generate(**batch, num_beams=4)
    with pytest.raises(TypeError):
        m.generate(**batch, not_an_option=1)
