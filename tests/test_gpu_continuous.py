"""GPU (-m gpu): continuous batching of regions in the decode loop (gar_amd/continuous.py) — rows retire at
their own EOS and queued regions are admitted into the running loop. Reference behaviour being served: one generate until EOS
per item with max_new_tokens = 1024 (evaluation/GAR-Bench/inference.py:158-170, demo/gar_with_mask.py:112-122)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tiny():
    from gar_amd import GARConfig
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.tiny()
    return cfg, synthetic_weights(cfg), GARProcessor.from_config(cfg, max_num_tiles=4)


def _sample(cfg, proc, i, w=200, h=160, dtype=torch.float32, multi=False):
    from gar_amd.eval_dataset import MultiRegionDataset, SingleRegionCaptionDataset
    from gar_amd.synthetic import synthetic_disjoint_masks, synthetic_image, synthetic_mask
    img = synthetic_image(i, w, h)
    if multi:
        masks = synthetic_disjoint_masks(i, 3, w, h)
        qs = "What is the relationship between <Prompt0>, <Prompt1> and <Prompt2>?"
        return MultiRegionDataset(img, masks, qs, proc, data_dtype=dtype, device="cpu")[0]
    return SingleRegionCaptionDataset(img, synthetic_mask(i, w, h), proc, data_dtype=dtype, device="cpu")[0]


def _pick_eos(streams, n_ids, max_new):
    """synthetic EOS ids: token ids out of the regions' own free-running streams, chosen greedily so that the captions they cut
    have MIXED lengths (some rows stop after a few tokens, some run to max_new_tokens)."""
    cand = sorted({t for s in streams for t in s})

    def lengths(eos):
        return [next((j + 1 for j, t in enumerate(s) if t in eos), max_new) for s in streams]
    eos = set()
    for _ in range(n_ids):
        best = None
        for c in cand:
            if c in eos:
                continue
            ls = lengths(eos | {c})
            spread = len(set(ls)) - 0.02 * abs(sum(ls) / len(ls) - 0.5 * max_new)
            if best is None or spread > best[0]:
                best = (spread, c)
        eos.add(best[1])
    return sorted(eos), lengths(eos)


def _single_runs(m, samples, max_new, eos):
    exp = []
    for s in samples:
        row = m.generate(**s, max_new_tokens=max_new, eos_token_id=eos).sequences[0].cpu().tolist()
        cut = next((j + 1 for j, t in enumerate(row) if t in eos), len(row))
        exp.append(row[:cut])
    return exp


@pytest.mark.parametrize("use_graph", [True, False])
def test_rows_retire_and_regions_are_admitted_inside_the_decode_loop(tiny, use_graph):
    """3 x B regions with captions of mixed lengths (EOS ids picked from the regions' own streams), B = 6 decode rows, f32:
    every region's tokens equal its single-region generate() run, and the number of decode steps executed is what the captions
    need — ceil(sum of decode tokens / B) + a drain tail — not the sum of the static batches' longest captions."""
    from gar_amd.continuous import ContinuousBatcher
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    B, NT = 6, 20
    m = GARModel(cfg, W, torch.float32)
    samples = [_sample(cfg, proc, 100 + i) for i in range(3 * B)]
    assert len({tuple(s["input_ids"].shape) for s in samples}) == 1
    streams = [m.generate(**s, max_new_tokens=NT).sequences[0].cpu().tolist() for s in samples]
    eos, lens = _pick_eos(streams, 6, NT)
    assert len(set(lens)) >= 4 and min(lens) <= NT // 3, lens          # mixed: short and long captions
    exp = _single_runs(m, samples, NT, eos)
    assert [len(e) for e in exp] == lens
    cb = ContinuousBatcher(m, slots=B, max_new_tokens=NT, eos_token_id=eos, poll_every=1, admit_min=1, use_graph=use_graph)
    tickets = [cb.submit(s) for s in samples]
    res = cb.flush()
    for t, e in zip(tickets, exp):
        assert res[t] == e, (t, res[t], e)
    st = cb.stats
    need = sum(n - 1 for n in lens)                      # decode tokens (the first token of a caption comes out of the prefill)
    static = sum(max(lens[i:i + B]) - 1 for i in range(0, len(lens), B))
    print(f"lengths {lens}; decode steps: continuous {st['decode_steps']}, static batches {static}, lower bound "
          f"{math.ceil(need / B)}; occupancy {st['live_row_steps'] / st['row_steps']:.2f}; {st['prompt_passes']} prompt passes")
    assert st["admitted"] == st["retired"] == 3 * B
    ones = sum(1 for n in lens if n == 1)            # a caption that IS its first token holds its row for one step before the poll
    assert math.ceil(need / B) <= st["decode_steps"] <= math.ceil((need + ones) / B) + max(lens)
    assert st["decode_steps"] < static
    assert st["live_row_steps"] >= need               # every needed token was decoded in a live row
    # the state is reusable: a second queue through the same batcher (same graph, rows recycled from a fresh base)
    t2 = [cb.submit(s) for s in samples[:B + 2]]
    res2 = cb.flush()
    assert [res2[t] for t in t2] == exp[:B + 2]


def test_admission_of_ragged_prompts_and_rebasing(tiny):
    """Prompts of DIFFERENT lengths (single-region and three-region questions) share the rows: a group is prefilled as one
    left-padded batch and every sequence is right-aligned at the shared clock; a small horizon forces the re-base (live rows
    shifted down inside the cache while they decode). Tokens equal the single-region runs (f32)."""
    from gar_amd.continuous import ContinuousBatcher
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    B, NT = 4, 40            # Smax = round_up(261 + 40, 64) = 320: a new row's life stops fitting 19 steps after the base
    m = GARModel(cfg, W, torch.float32)
    samples = [_sample(cfg, proc, 200 + i, multi=(i % 3 == 1)) for i in range(16)]
    assert len({int(s["input_ids"].shape[1]) for s in samples}) >= 2
    streams = [m.generate(**s, max_new_tokens=NT).sequences[0].cpu().tolist() for s in samples]
    eos, lens = _pick_eos(streams, 5, NT)
    exp = _single_runs(m, samples, NT, eos)
    cb = ContinuousBatcher(m, slots=B, max_new_tokens=NT, eos_token_id=eos, poll_every=2, admit_min=1, horizon=0, smax_multiple=64)
    tickets = [cb.submit(s) for s in samples]
    res = cb.flush()
    for t, e in zip(tickets, exp):
        assert res[t] == e, (t, res[t], e)
    print(f"lengths {lens}; stats {cb.stats}; Smax {cb.Smax}")
    assert cb.stats["rebases"] >= 1, cb.stats


def test_generate_eos_latches_on_the_device(tiny):
    """generate(eos_token_id=[...]) stops on the device-side latch (gar_argmax finished / done_count): the padded output is what
    the host scan of the token matrix produced before (rows cut at their own eos, pad behind it, batch cut at the last row)."""
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    m = GARModel(cfg, W, torch.float32)
    ss = [_sample(cfg, proc, 300 + i) for i in range(5)]
    batch = dict(input_ids=torch.cat([s["input_ids"] for s in ss]), pixel_values=torch.cat([s["pixel_values"] for s in ss]),
                 global_mask_values=torch.cat([s["global_mask_values"] for s in ss]), bboxes=[s["bboxes"][0] for s in ss],
                 aspect_ratios=torch.cat([s["aspect_ratios"] for s in ss]))
    NT = 16
    free = m.generate(**batch, max_new_tokens=NT).sequences.cpu().tolist()
    eos, lens = _pick_eos(free, 3, NT)
    pad = 7
    for sync_every in (1, 4, 16):
        out = m.generate(**batch, max_new_tokens=NT, eos_token_id=eos, sync_every=sync_every,
                         generation_config=None).sequences.cpu().tolist()
        cut = max(lens)
        for b in range(5):
            want = free[b][:lens[b]] + [eos[0]] * (cut - lens[b])
            assert out[b][:cut] == want, (sync_every, b)
        assert len(out[0]) >= cut
    out = m.generate(**batch, generation_config=dict(max_new_tokens=NT, eos_token_id=eos, pad_token_id=pad, do_sample=False))
    rows = out.sequences.cpu().tolist()
    assert all(rows[b][:max(lens)] == free[b][:lens[b]] + [pad] * (max(lens) - lens[b]) for b in range(5))
    assert len(rows[0]) == max(lens) or max(lens) == NT
    # a list longer than the device table keeps the host scan: same output
    seen = {t for r in free for t in r}
    many = eos + [t for t in range(cfg.mllm_config.text_config.vocab_size) if t not in seen][:m.MAX_EOS_IDS]
    assert len(many) > m.MAX_EOS_IDS
    out2 = m.generate(**batch, max_new_tokens=NT, eos_token_id=many, generation_config=dict(pad_token_id=pad))
    assert out2.sequences.cpu().tolist() == rows


def test_continuous_at_gar1b_dimensions_f32():
    """GAR-1B's real dimensions (17 tiles of 1025 tokens, S ~ 4.7k, head_dim 64, GQA 32 / 8) with one ViT layer and two Llama
    layers, f32: ten regions through four decode rows — admissions at cache offsets that are not multiples of a kv tile — give
    every region the tokens of its own single-region generate()."""
    from gar_amd import GARConfig
    from gar_amd.continuous import ContinuousBatcher
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    m = GARModel(cfg, W, torch.float32)
    B, NT = 4, 14
    samples = [_sample(cfg, proc, 400 + i, 1024, 1024) for i in range(10)]
    assert samples[0]["pixel_values"].shape[0] == 17 and samples[0]["input_ids"].shape[1] > 4600
    streams = [m.generate(**s, max_new_tokens=NT).sequences[0].cpu().tolist() for s in samples]
    eos, lens = _pick_eos(streams, 4, NT)
    exp = _single_runs(m, samples, NT, eos)
    cb = ContinuousBatcher(m, slots=B, max_new_tokens=NT, eos_token_id=eos, poll_every=3, admit_min=1)
    tickets = [cb.submit(s) for s in samples]
    res = cb.flush()
    print(f"lengths {lens}; stats {cb.stats}")
    for t, e in zip(tickets, exp):
        assert res[t] == e, (t, res[t], e)
    assert cb.stats["prompt_passes"] >= 3 and cb.stats["decode_steps"] < sum(max(lens[i:i + B]) - 1 for i in range(0, 10, B)) + NT


def test_two_batchers_on_one_model_do_not_share_state(tiny):
    """every ContinuousBatcher owns its KV-state slots of the model — two batchers with the same (slots, Smax), pumped
    alternately, give each region the tokens of its single-region run (they used to alias one cache, token log and graph)."""
    from gar_amd.continuous import ContinuousBatcher
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    B, NT = 4, 12
    m = GARModel(cfg, W, torch.float32)
    sa = [_sample(cfg, proc, 300 + i) for i in range(2 * B)]
    sb = [_sample(cfg, proc, 400 + i) for i in range(2 * B)]
    streams = [m.generate(**s, max_new_tokens=NT).sequences[0].cpu().tolist() for s in sa + sb]
    eos, lens = _pick_eos(streams, 4, NT)
    exp = _single_runs(m, sa + sb, NT, eos)
    a = ContinuousBatcher(m, slots=B, max_new_tokens=NT, eos_token_id=eos, poll_every=1, admit_min=1)
    b = ContinuousBatcher(m, slots=B, max_new_tokens=NT, eos_token_id=eos, poll_every=1, admit_min=1)
    assert {a.DEC_SLOT, a.STAGE_SLOT}.isdisjoint({b.DEC_SLOT, b.STAGE_SLOT}) and min(a.DEC_SLOT, b.DEC_SLOT) >= 2
    ta = [a.submit(s) for s in sa]
    tb = [b.submit(s) for s in sb]
    for _ in range(400):                       # alternate single cycles of the two loops
        if a.queue or a.n_active:
            a._cycle(True)
        if b.queue or b.n_active:
            b._cycle(True)
        if not (a.queue or a.n_active or b.queue or b.n_active):
            break
    assert [a.results[t] for t in ta] == exp[:2 * B]
    assert [b.results[t] for t in tb] == exp[2 * B:]
