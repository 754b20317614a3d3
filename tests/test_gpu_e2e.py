"""GPU (-m gpu): GARModel.generate on the HIP path against the CPU oracle (token-for-token greedy parity in f32
mode, stated tolerances on logits; bf16 mode measured against the f32 oracle), plus size-independent properties at
the BASELINE.json sizes."""
import pytest
import torch

from parity_util import F32_LOGIT_TOL, assert_discriminating      # 2e-4 * max|logit|; margins >= 10 x that

pytestmark = pytest.mark.gpu

BF16_FEAT_TOL = 3e-2      # relative L2 error of image features in bf16 mode vs the f32 oracle
FP16_FEAT_TOL, FP16_LOGIT_TOL = 4e-3, 8e-3     # the same two bounds for the fp16 twin library (three more mantissa bits)
BF16_LOGIT_TOL = 4e-2     # ... of first-token logits (two more layers of bf16 rounding on top of the features)
# sample indices whose oracle sequences are asserted discriminating on the CPU (tests/test_parity_evidence.py)
TINY_SINGLE, TINY_MULTI, TINY_VIDEO_BASE = 0, 0, 50
TINY8B_SINGLE, TINY8B_VIDEO_BASE = 1, 70


# bf16 vs f32 at a teacher-forced step of the TINY model (decode-schedule tests): the bound the full-depth tests used before round 5
TINY_BF16_STEP_REL_L2 = 6e-2

def _sample(cfg, proc, i=0, w=200, h=160, dtype=torch.float32, multi=False):
    from gar_amd.eval_dataset import MultiRegionDataset, SingleRegionCaptionDataset
    from gar_amd.synthetic import synthetic_disjoint_masks, synthetic_image, synthetic_mask
    img = synthetic_image(i, w, h)
    if multi:
        masks = synthetic_disjoint_masks(i, 3, w, h)
        qs = "What is the relationship between <Prompt0>, <Prompt1> and <Prompt2>?"
        return MultiRegionDataset(img, masks, qs, proc, data_dtype=dtype, device="cpu",
                                  prompt_order=["<Prompt0>", "<Prompt2>", "<Prompt1>"])[0]
    return SingleRegionCaptionDataset(img, synthetic_mask(i, w, h), proc, data_dtype=dtype, device="cpu")[0]


def _oracle(W, cfg, s, n, **kw):
    from oracle import gar_oracle as O
    return O.gar_generate(W, cfg, s["pixel_values"].float(), s["global_mask_values"].float(), s["aspect_ratios"],
                          s["bboxes"], s["input_ids"], None, max_new_tokens=n, return_logits=True, **kw)


def _rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / b.norm())


@pytest.fixture(scope="module")
def tiny():
    from gar_amd import GARConfig
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.tiny()
    return cfg, synthetic_weights(cfg), GARProcessor.from_config(cfg, max_num_tiles=4)


def _check_f32(out, ref_seq, ref_logits, what=""):
    """token-for-token greedy parity + the stated logit tolerance, every decode step."""
    assert out.sequences.cpu().tolist() == ref_seq.tolist(), what
    err = float((out.logits.cpu() - ref_logits).abs().max())
    assert err <= F32_LOGIT_TOL * float(ref_logits.abs().max()), (what, err)


@pytest.mark.parametrize("multi", [False, True])
def test_f32_greedy_parity_tiny(tiny, multi):
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    s = _sample(cfg, proc, TINY_MULTI if multi else TINY_SINGLE, multi=multi)
    ref_seq, ref_logits = _oracle(W, cfg, s, 12)
    assert_discriminating(ref_seq, ref_logits)
    m = GARModel(cfg, W, torch.float32)
    # eager launches and the hipGraph-replayed decode loop (the path bench.py times): tokens AND per-step logits
    _check_f32(m.generate(**s, max_new_tokens=12, return_logits=True, use_graph=False), ref_seq, ref_logits, "eager")
    _check_f32(m.generate(**s, max_new_tokens=12, return_logits=True, use_graph=True), ref_seq, ref_logits, "graph")
    # a second request on the cached graph (other tokens in the buffers) and without logit read-back
    _check_f32(m.generate(**s, max_new_tokens=12, return_logits=True), ref_seq, ref_logits, "graph replayed")
    assert m.generate(**s, max_new_tokens=12).sequences.cpu().tolist() == ref_seq.tolist()


def _video_sample(cfg, proc, n_frames=3, dtype=torch.float32, base=50, w=180, h=150):
    from gar_amd.eval_dataset import VideoRegionCaptionDataset
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    frames = [synthetic_image(base + f, w, h) for f in range(n_frames)]
    masks = [synthetic_mask(base + 10 + f, w, h) for f in range(n_frames)]
    return VideoRegionCaptionDataset(frames, masks, proc, data_dtype=dtype, device="cpu")[0]


def test_f32_video_replay_parity_tiny(tiny):
    """A13: per-frame crop tokens, one P x P map per frame, no thumbnail (modeling_perception_lm.py:765-852)."""
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    s = _video_sample(cfg, proc, 3, base=TINY_VIDEO_BASE)
    assert s["pixel_values"].shape[0] == 3 and len(s["bboxes"][0]) == 3
    ref_seq, ref_logits = O.gar_generate(W, cfg, s["pixel_values"], s["global_mask_values"], None, s["bboxes"],
                                         s["input_ids"], None, max_new_tokens=8, return_logits=True,
                                         video_frame_tokens=s["video_frame_tokens"])
    assert_discriminating(ref_seq, ref_logits)
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True, use_graph=False), ref_seq, ref_logits, "eager")
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True), ref_seq, ref_logits, "graph")
    # the replayed rows equal the oracle's (and differ from the image-path replay of the same inputs)
    feats = m.get_image_features(s["pixel_values"], s["global_mask_values"])
    emb = m.build_inputs_embeds(s["input_ids"], feats, s["bboxes"], None, 3, True, s["video_frame_tokens"]).cpu().clone()
    ref_emb = O.build_inputs_embeds(W, cfg, s["pixel_values"], s["global_mask_values"], None, s["bboxes"], s["input_ids"],
                                    video_frame_tokens=s["video_frame_tokens"])
    assert _rel_l2(emb, ref_emb) < 1e-5


def test_f32_intermediates_tiny(tiny):
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 2)
    m = GARModel(cfg, W, torch.float32)
    ref_emb, inter = O.build_inputs_embeds(W, cfg, s["pixel_values"], s["global_mask_values"], s["aspect_ratios"],
                                           s["bboxes"], s["input_ids"], return_intermediates=True)
    feats = m.get_image_features(s["pixel_values"], s["global_mask_values"])
    assert _rel_l2(feats, inter["image_features"]) < 1e-5
    emb = m.build_inputs_embeds(s["input_ids"], feats, s["bboxes"], s["aspect_ratios"], s["pixel_values"].shape[0])
    assert _rel_l2(emb, ref_emb) < 1e-5
    # rows that are plain token embeddings are bit-exact copies
    ids = s["input_ids"][0]
    plain = (ids != cfg.mllm_config.image_token_id) & ~torch.isin(ids, torch.tensor(cfg.crop_tokens_ids))
    assert torch.equal(emb[0].cpu()[plain], ref_emb[0][plain])


def test_mllm_helper_api(tiny):
    """`model.mllm.*` / `get_input_embeddings` of the reference (modeling_gar.py:64-72,332-346) for callers that script
    the stages of generate() themselves."""
    from gar_amd import hip
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 2)
    m = GARModel(cfg, W, torch.float32)
    emb = m.mllm.get_input_embeddings()(s["input_ids"])
    ref = torch.nn.functional.embedding(s["input_ids"], W[O.LM + "embed_tokens.weight"])
    assert torch.equal(emb.cpu(), ref) and torch.equal(m.get_input_embeddings()(s["input_ids"][0, :7]).cpu(), ref[0, :7])
    feats = m.mllm.get_image_features(s["pixel_values"], global_mask_values=s["global_mask_values"]).clone()
    img_mask, vid_mask = m.mllm.get_placeholder_mask(s["input_ids"], emb, image_features=feats)
    want = (s["input_ids"] == cfg.mllm_config.image_token_id).unsqueeze(-1).expand_as(ref)
    assert torch.equal(img_mask.cpu(), want) and not bool(vid_mask.any())
    # masked_scatter with that mask is the oracle's embed_and_scatter (modeling_gar.py:341-346)
    scattered = emb.masked_scatter(img_mask, feats)
    want_emb = O.embed_and_scatter(s["input_ids"], W[O.LM + "embed_tokens.weight"], feats.cpu(), cfg.mllm_config.image_token_id)
    assert torch.equal(scattered.cpu(), want_emb)
    with pytest.raises(ValueError, match="do not match"):
        m.mllm.get_placeholder_mask(s["input_ids"], emb, image_features=feats[:-1])
    with pytest.raises(hip.GarError, match="either mask_embeds"):
        m.mllm.get_image_features(s["pixel_values"], mask_embeds=torch.zeros(1), global_mask_values=s["global_mask_values"])


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_reference_generate_prologue_through_the_facade(tiny, dt):
    """The body of the reference's GARModel.generate up to the feature replay (modeling_gar.py:315-346), statement by
    statement, against THIS model's sub-API — mask decode in torch as the reference writes it, `model.mask_patch_embedding`,
    `mllm.get_input_embeddings()`, `mllm.get_image_features(pixel_values=, mask_embeds=)`, `mllm.get_placeholder_mask`,
    `masked_scatter` — equals what generate() builds internally (build_inputs_embeds before the replay rows are overwritten:
    compared on the rows the replay does not touch) and the oracle's mask embeddings."""
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 2, dtype=dt)
    self = GARModel(cfg, W, dt)
    device = self.device
    pixel_values, global_mask_values, input_ids = s["pixel_values"], s["global_mask_values"], s["input_ids"]
    # ---- reference lines 315-346 ------------------------------------------------------------------------------------
    pixel_values = pixel_values.to(device).to(self.mllm.dtype)
    mask_values = torch.round((global_mask_values + 1.0) / 2.0 * 255.0).long().to(device)
    mask_values = torch.clamp(mask_values, min=0, max=self.prompt_numbers)
    assert mask_values.max() < self.prompt_numbers + 1 and mask_values.min() >= 0
    mask_embeds = self.mask_patch_embedding((mask_values != self.prompt_numbers).to(self.mllm.dtype))
    inputs_embeds = self.mllm.get_input_embeddings()(input_ids)
    image_features = self.mllm.get_image_features(pixel_values=pixel_values, mask_embeds=mask_embeds)
    image_features = image_features.to(inputs_embeds.device, dtype=inputs_embeds.dtype)
    special_image_mask, _ = self.mllm.get_placeholder_mask(input_ids, inputs_embeds=inputs_embeds, image_features=image_features)
    inputs_embeds = inputs_embeds.masked_scatter(special_image_mask, image_features)
    # -----------------------------------------------------------------------------------------------------------------
    v = cfg.mllm_config.vision_config
    T = pixel_values.shape[0]
    assert tuple(mask_embeds.shape) == (T, v.embed_dim, v.grid, v.grid)
    Wd = {k: t.to(dt).float() for k, t in W.items()}
    me_ref = O.mask_patch_embed(O.decode_mask_values(global_mask_values.float(), cfg.prompt_numbers), Wd["mask_patch_embedding.weight"])
    tol = 1e-5 if dt == torch.float32 else 1.5e-2
    assert _rel_l2(mask_embeds.float().cpu(), me_ref) < tol
    # the fused form generate() runs (mask conv as K columns of the patch-embed GEMM)
    fused_feats = self.get_image_features(s["pixel_values"], s["global_mask_values"]).clone()
    assert _rel_l2(image_features, fused_feats) < (1e-5 if dt == torch.float32 else BF16_FEAT_TOL)
    feats_ref = O.get_image_features(s["pixel_values"].float(), me_ref, Wd, cfg)
    assert _rel_l2(image_features.float().cpu(), feats_ref) < (1e-4 if dt == torch.float32 else BF16_FEAT_TOL)
    built = self.build_inputs_embeds(input_ids, fused_feats, s["bboxes"], s["aspect_ratios"], T)
    crop = torch.zeros(input_ids.shape[1], dtype=torch.bool)
    for tok in cfg.crop_tokens_ids:
        crop |= (input_ids[0] == tok)
    keep = ~crop
    assert _rel_l2(inputs_embeds[0, keep.to(device)], built[0, keep.to(device)]) < (1e-5 if dt == torch.float32 else BF16_FEAT_TOL)
    assert torch.equal(self.mask_patch_embedding.weight.float().cpu(), W["mask_patch_embedding.weight"].to(dt).float())


def test_generation_options_that_would_change_greedy_tokens_are_refused(tiny):
    """modeling_gar.py:418-426 forwards any GenerationConfig to HF; here greedy search and temperature / top-k / top-p sampling
    exist (tests/test_gpu_sampling.py): beams / penalties / the other warpers raise (they used to be ignored), sampling-only knobs
    without do_sample are accepted as HF accepts them."""
    from gar_amd import hip
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 2)
    m = GARModel(cfg, W, torch.float32)
    base = m.generate(**s, generation_config=dict(max_new_tokens=3, do_sample=False, num_beams=1, repetition_penalty=1.0,
                                                  temperature=0.7, top_p=0.9))
    assert base.sequences.shape == (1, 3)
    for bad in (dict(num_beams=4), dict(repetition_penalty=1.2), dict(no_repeat_ngram_size=2), dict(min_new_tokens=5),
                dict(bad_words_ids=[[3]]), dict(do_sample=True, min_p=0.1), dict(do_sample=True, num_beams=2)):
        with pytest.raises(hip.GarError):
            m.generate(**s, generation_config=dict(max_new_tokens=3, **bad))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pruned_last_prefill_layer_equals_full_width(tiny, dt):
    """PRUNE_LAST_PREFILL_LAYER: the last Llama layer of a prefill runs attention / o / gate-up / down for the last prompt row
    only (what lm_head reads, modeling_perception_lm.py:545-552). Same tokens and logits as the full-width layer up to the
    summation order of a different GEMM kernel; the KV caches — every row, all layers — are bit-identical (the qkv GEMM still
    runs over all rows). Ragged (left-padded) batch included."""
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    ss = [_sample(cfg, proc, i, dtype=dt) for i in (2, 3, 5)]
    batch = dict(input_ids=torch.cat([x["input_ids"] for x in ss]), pixel_values=torch.cat([x["pixel_values"] for x in ss]),
                 global_mask_values=torch.cat([x["global_mask_values"] for x in ss]), bboxes=[x["bboxes"][0] for x in ss],
                 aspect_ratios=torch.cat([x["aspect_ratios"] for x in ss]))
    m = GARModel(cfg, W, dt)
    assert m.PRUNE_LAST_PREFILL_LAYER
    n = 6
    pruned = m.generate(**batch, max_new_tokens=n, return_logits=True)
    st = m._ws[m._llm_lru[-1]]
    kc, vc = st["Kc"].clone(), st["Vc"].clone()
    assert "att_last" in m._ws[("prefill",)]
    m.PRUNE_LAST_PREFILL_LAYER = False
    full = m.generate(**batch, max_new_tokens=n, return_logits=True, forced_tokens=pruned.sequences)
    st = m._ws[m._llm_lru[-1]]
    S = batch["input_ids"].shape[1]
    assert torch.equal(kc[:, :, :, :S], st["Kc"][:, :, :, :S]) and torch.equal(vc[:, :, :, :S], st["Vc"][:, :, :, :S])
    tol = 2e-5 if dt == torch.float32 else BF16_LOGIT_TOL
    for j in range(n):
        assert _rel_l2(pruned.logits[:, j], full.logits[:, j]) < tol, j
    if dt == torch.float32:
        assert torch.equal(pruned.sequences, full.sequences)


def test_validate_false_flags_a_right_padded_attention_mask(tiny):
    """generate(validate=False) takes left_pad from the mask's zero count without a host sync; gar_input_check looks at the
    mask's FORM on the device: a row that is not 0...01...1 sets INPUT_MASK_NOT_LEFT_PADDED (validate=True raises on the host)."""
    from gar_amd import hip
    from gar_amd.modeling_gar import GARModel, INPUT_MASK_NOT_LEFT_PADDED
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 2)
    m = GARModel(cfg, W, torch.float32)
    S = s["input_ids"].shape[1]
    ok = torch.ones(1, S, dtype=torch.int64)
    out = m.generate(**{**s, "attention_mask": ok}, max_new_tokens=2, validate=False)
    assert int(out.input_flags.item()) == 0
    right = ok.clone()
    right[0, -3:] = 0
    out = m.generate(**{**s, "attention_mask": right}, max_new_tokens=2, validate=False)
    assert int(out.input_flags.item()) & INPUT_MASK_NOT_LEFT_PADDED
    hole = ok.clone()
    hole[0, 5] = 0                       # 1 1 1 1 1 0 1 ...: not a prefix of zeros
    out = m.generate(**{**s, "attention_mask": hole}, max_new_tokens=2, validate=False)
    assert int(out.input_flags.item()) & INPUT_MASK_NOT_LEFT_PADDED
    with pytest.raises(hip.GarError, match="LEFT-padded"):
        m.generate(**{**s, "attention_mask": right}, max_new_tokens=2)


def test_batch_equals_singles_f32(tiny):
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    a, b = _sample(cfg, proc, 3), _sample(cfg, proc, 4)
    assert a["input_ids"].shape == b["input_ids"].shape
    m = GARModel(cfg, W, torch.float32)
    sa = m.generate(**a, max_new_tokens=6).sequences.cpu()
    sb = m.generate(**b, max_new_tokens=6).sequences.cpu()
    both = dict(input_ids=torch.cat([a["input_ids"], b["input_ids"]]),
                pixel_values=torch.cat([a["pixel_values"], b["pixel_values"]]),
                global_mask_values=torch.cat([a["global_mask_values"], b["global_mask_values"]]),
                bboxes=a["bboxes"] + b["bboxes"], aspect_ratios=torch.cat([a["aspect_ratios"], b["aspect_ratios"]]))
    sab = m.generate(**both, max_new_tokens=6).sequences.cpu()
    assert torch.equal(sab, torch.cat([sa, sb]))


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_planned_passes_equal_region_chunks(tiny, dt):
    """The vision tower over chunks of image TILES and the prefill over chunks of SEQUENCES (gar_amd/planner.py; with
    small row caps so that three samples really split: tiles cut inside a sample) give the same tokens and logits as
    both passes over one region at a time (prefill_chunk=1) — a row's result does not depend on the pass it rides in."""
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    ss = [_sample(cfg, proc, 3 + i, dtype=dt) for i in range(3)]
    assert len({tuple(x["input_ids"].shape) for x in ss}) == 1
    batch = dict(input_ids=torch.cat([x["input_ids"] for x in ss]), pixel_values=torch.cat([x["pixel_values"] for x in ss]),
                 global_mask_values=torch.cat([x["global_mask_values"] for x in ss]),
                 bboxes=sum((x["bboxes"] for x in ss), []), aspect_ratios=torch.cat([x["aspect_ratios"] for x in ss]))
    m1 = GARModel(cfg, W, dt, prefill_chunk=1)
    ref = m1.generate(**batch, max_new_tokens=5, return_logits=True)
    m2 = GARModel(cfg, W, dt)
    tiles = batch["pixel_values"].shape[0] // 3
    v = cfg.mllm_config.vision_config
    m2.VIT_CHUNK_ROWS = 2 * (v.num_patches + m2.npt) * max(1, tiles // 2)       # two thirds of a sample's tiles per pass
    m2.PREFILL_CHUNK_ROWS = 2 * batch["input_ids"].shape[1]                      # two sequences per pass
    tc, sc = m2._plan_passes(3, tiles, batch["input_ids"].shape[1])
    assert sum(tc) == 3 * tiles and sum(sc) == 3 and len(sc) >= 2 and any(c % tiles for c in tc)
    out = m2.generate(**batch, max_new_tokens=5, return_logits=True)
    assert torch.equal(out.sequences, ref.sequences)
    assert torch.equal(out.logits, ref.logits)


def test_bf16_against_f32_oracle_tiny(tiny):
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 5, dtype=torch.bfloat16)
    Wq = {k: v.to(torch.bfloat16).float() for k, v in W.items()}          # the oracle sees the bf16-rounded weights
    ref_seq, ref_logits = _oracle(Wq, cfg, s, 8)
    m = GARModel(cfg, W, torch.bfloat16)
    out = m.generate(**s, max_new_tokens=8, return_logits=True)
    lg = out.logits.cpu()
    assert _rel_l2(lg[:, 0], ref_logits[:, 0]) < BF16_LOGIT_TOL
    # top-1 agreement wherever the f32 margin exceeds the measured bf16 logit error
    err = float((lg[:, 0] - ref_logits[:, 0]).abs().max())
    top2 = ref_logits[:, 0].topk(2).values
    if float(top2[0, 0] - top2[0, 1]) > 2 * err:
        assert int(out.sequences[0, 0]) == int(ref_seq[0, 0])
    feats = m.get_image_features(s["pixel_values"], s["global_mask_values"])
    _, inter = O.build_inputs_embeds(Wq, cfg, s["pixel_values"].float(), s["global_mask_values"].float(),
                                     s["aspect_ratios"], s["bboxes"], s["input_ids"], return_intermediates=True)
    assert _rel_l2(feats, inter["image_features"]) < BF16_FEAT_TOL
    # graph replay == eager, bit for bit (same kernels, same order)
    g = m.generate(**s, max_new_tokens=8)
    assert torch.equal(g.sequences.cpu(), out.sequences.cpu())


@pytest.mark.parametrize("dims", ["tiny", "gar_1b_two_layers"])
def test_fp16_against_f32_oracle(tiny, dims):
    """--data_type fp16 of the reference CLIs (demo/gar_with_mask.py:41-45): the same kernels with IEEE binary16 as the 16-bit
    element type (libgar_hip_f16.so, hip.lib(torch.float16)) against the f32 oracle on the fp16-rounded weights — tiny
    dimensions and GAR-1B dimensions (two ViT blocks + two Llama layers: folded norms, tile GEMM epilogues, head_dim 64
    attention with the lazy max whose stored softmax weights must stay below fp16's 65504)."""
    from gar_amd import GARConfig, hip
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    from oracle import gar_oracle as O
    dt = torch.float16
    if dims == "tiny":
        cfg, W, proc = tiny
        s, n, kw = _sample(cfg, proc, 5, dtype=dt), 8, {}
    else:
        cfg = GARConfig.gar_1b(**{"vision.depth": 2, "text.num_hidden_layers": 2})
        W, proc = synthetic_weights(cfg), GARProcessor.from_config(cfg, max_num_tiles=16)
        s, n, kw = _sample(cfg, proc, 3, 1024, 1024, dtype=dt), 4, dict(attn_impl="sdpa")
    Wq = {k: v.to(dt).float() for k, v in W.items()}
    ref_seq, ref_logits = _oracle(Wq, cfg, s, n, **kw)
    m = GARModel(cfg, W, dt)
    out = m.generate(**s, max_new_tokens=n, return_logits=True)
    assert hip._lib_f16 is not None                                    # the twin library served it
    lg = out.logits.cpu()
    assert torch.isfinite(lg).all()
    assert _rel_l2(lg[:, 0], ref_logits[:, 0]) < FP16_LOGIT_TOL
    err = float((lg[:, 0] - ref_logits[:, 0]).abs().max())
    top2 = ref_logits[:, 0].topk(2).values
    if float(top2[0, 0] - top2[0, 1]) > 2 * err:
        assert int(out.sequences[0, 0]) == int(ref_seq[0, 0])
    feats = m.get_image_features(s["pixel_values"], s["global_mask_values"])
    _, inter = O.build_inputs_embeds(Wq, cfg, s["pixel_values"].float(), s["global_mask_values"].float(),
                                     s["aspect_ratios"], s["bboxes"], s["input_ids"], return_intermediates=True)
    assert _rel_l2(feats, inter["image_features"]) < FP16_FEAT_TOL
    g = m.generate(**s, max_new_tokens=n)                               # graph replay == eager
    assert torch.equal(g.sequences.cpu(), out.sequences.cpu())


def test_eos_stops_and_pads(tiny):
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 1)
    m = GARModel(cfg, W, torch.float32)
    full = m.generate(**s, max_new_tokens=10).sequences[0].tolist()
    eos = full[3]
    first = full.index(eos)
    cut = m.generate(**s, max_new_tokens=10, eos_token_id=eos, sync_every=2).sequences[0].tolist()
    assert cut == full[:first + 1]


def test_reference_error_behaviour(tiny):
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 1)
    m = GARModel(cfg, W, torch.float32)
    bad = dict(s)
    bad["input_ids"] = s["input_ids"].clone()
    bad["input_ids"][0, 10] = 7                      # one image placeholder fewer than features
    with pytest.raises(ValueError, match="Image features and image tokens do not match"):
        m.generate(**bad, max_new_tokens=2)
    bad = dict(s)
    bad["bboxes"] = [{}]
    with pytest.raises(KeyError):
        m.generate(**bad, max_new_tokens=2)
    # validate=False skips the host syncs, not the checks: the same conditions are evaluated on the device and come
    # back as a flag (raised at the first EOS poll when there is one)
    from gar_amd.modeling_gar import INPUT_COUNT_MISMATCH, INPUT_ID_RANGE, INPUT_MISSING_BBOX, INPUT_SPAN_LENGTH
    assert int(m.generate(**s, max_new_tokens=2, validate=False).input_flags.item()) == 0
    assert m.generate(**s, max_new_tokens=2).input_flags is None
    bad = dict(s)
    bad["input_ids"] = s["input_ids"].clone()
    bad["input_ids"][0, 10] = 7
    assert int(m.generate(**bad, max_new_tokens=2, validate=False).input_flags.item()) == INPUT_COUNT_MISMATCH
    with pytest.raises(ValueError, match="image token count"):
        m.generate(**bad, max_new_tokens=4, validate=False, eos_token_id=0, sync_every=2)
    bad = dict(s)
    bad["bboxes"] = [{}]
    assert int(m.generate(**bad, max_new_tokens=2, validate=False).input_flags.item()) == INPUT_MISSING_BBOX
    crop = next(c for c in cfg.crop_tokens_ids if bool((s["input_ids"][0] == c).any()))
    pos = (s["input_ids"][0] == crop).nonzero()[:, 0]
    bad = dict(s)
    bad["input_ids"] = s["input_ids"].clone()
    bad["input_ids"][0, pos[-1]] = 7                  # the crop-token span is one row short
    assert int(m.generate(**bad, max_new_tokens=2, validate=False).input_flags.item()) == INPUT_SPAN_LENGTH
    bad = dict(s)
    bad["input_ids"] = s["input_ids"].clone()
    bad["input_ids"][0, 0] = cfg.mllm_config.text_config.vocab_size + 5
    assert int(m.generate(**bad, max_new_tokens=2, validate=False).input_flags.item()) == INPUT_ID_RANGE


def test_f32_parity_gar1b_dims_one_layer():
    """GAR-1B shapes (1024 px-wide ViT, 2048-wide Llama, 128262 vocab, 17 tiles of a 1024^2 image, S ~ 4.7k) with one ViT
    and two Llama layers: exercises the exact GEMM / attention shapes of the benchmark config against the f32 oracle,
    8 greedy tokens, eager and hipGraph decode, logits at every step."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    s = _sample(cfg, proc, 3, 1024, 1024)
    assert s["pixel_values"].shape[0] == 17
    ref_seq, ref_logits = _oracle(W, cfg, s, 8, attn_impl="sdpa")
    assert_discriminating(ref_seq, ref_logits)
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True), ref_seq, ref_logits, "graph")
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True, use_graph=False), ref_seq, ref_logits, "eager")


def test_full_size_bf16_properties():
    """Full GAR-1B (23 + 16 layers) bf16 at the benchmark shape: finite, deterministic, graph == eager,
    batch of two identical samples yields identical rows."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    cfg = GARConfig.gar_1b()
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    s = _sample(cfg, proc, 0, 1024, 1024, dtype=torch.bfloat16)
    m = GARModel.from_synthetic(cfg, 0, torch.bfloat16)
    a = m.generate(**s, max_new_tokens=8, return_logits=True)
    assert torch.isfinite(a.logits).all()
    b = m.generate(**s, max_new_tokens=8)
    assert torch.equal(a.sequences, b.sequences)
    two = dict(input_ids=torch.cat([s["input_ids"]] * 2), pixel_values=torch.cat([s["pixel_values"]] * 2),
               global_mask_values=torch.cat([s["global_mask_values"]] * 2), bboxes=s["bboxes"] * 2,
               aspect_ratios=torch.cat([s["aspect_ratios"]] * 2))
    c = m.generate(**two, max_new_tokens=8)
    # the two rows of one batch run through the same kernels in the same order; a batch of two and a batch of one do not
    # (the split-KV decode attention divides the kv range by the batch size): in bf16 their logits differ by rounding
    # and a caption may legitimately fork at a low-margin step, so only the first token is compared across batch sizes
    assert torch.equal(c.sequences[0], c.sequences[1]) and int(c.sequences[0, 0]) == int(a.sequences[0, 0])


def test_bf16_decode_above_16_sequences_matches_f32_teacher_forced(tiny):
    """More than 16 sequences per step take the other decode schedule (separate RMSNorm launches, `down` as split-K slices
    reduced by gar_splitk_residual_rmsnorm, single-split attention writing its output itself): 20 sequences in bf16,
    teacher-forced on the f32 HIP run of the same batch — per-step logits within the bf16 tolerance, repeated samples give
    identical rows, hipGraph replay == eager launches."""
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    ss = [_sample(cfg, proc, i) for i in (3, 4, 6, 7)]
    assert len({tuple(s["input_ids"].shape) for s in ss}) == 1
    order = [0, 1, 2, 3, 0] * 4                                   # 20 rows, sample 0 at rows 0, 4, 5, 9, ...

    def batch(dt):
        return dict(input_ids=torch.cat([ss[i]["input_ids"] for i in order]),
                    pixel_values=torch.cat([ss[i]["pixel_values"] for i in order]).to(dt),
                    global_mask_values=torch.cat([ss[i]["global_mask_values"] for i in order]).to(dt),
                    bboxes=[ss[i]["bboxes"][0] for i in order],
                    aspect_ratios=torch.cat([ss[i]["aspect_ratios"] for i in order]))
    n = 6
    m32, m16 = GARModel(cfg, W, torch.float32), GARModel(cfg, W, torch.bfloat16)
    r32 = m32.generate(**batch(torch.float32), max_new_tokens=n, return_logits=True)
    r16 = m16.generate(**batch(torch.bfloat16), max_new_tokens=n, return_logits=True, forced_tokens=r32.sequences)
    assert m16.DOWN_SPLIT_K != 1 and ("decode", 20) in m16._ws and "down_partial" in m16._ws[("decode", 20)]
    m16u = GARModel(cfg, W, torch.bfloat16)
    m16u.DOWN_SPLIT_K = 1                                         # the unsplit schedule: gemm(EPI_RES) + rmsnorm
    u16 = m16u.generate(**batch(torch.bfloat16), max_new_tokens=n, return_logits=True, forced_tokens=r32.sequences)
    assert "down_partial" not in m16u._ws[("decode", 20)]
    for j in range(n):
        # bf16 vs f32 at a teacher-forced step (the first-token tolerance + what later steps add, as in the full-depth test)
        assert _rel_l2(r16.logits[:, j], r32.logits[:, j]) < TINY_BF16_STEP_REL_L2, j
        assert _rel_l2(u16.logits[:, j], r32.logits[:, j]) < TINY_BF16_STEP_REL_L2, j
        # split vs unsplit differ by the fp32 summation order of `down` only: an order of magnitude closer to each other
        assert _rel_l2(r16.logits[:, j], u16.logits[:, j]) < 5e-3, j
    lg = r16.logits.cpu()
    for a, b in ((0, 4), (0, 5), (1, 6), (3, 8)):                 # rows of the same sample
        assert order[a] == order[b] and torch.equal(lg[a], lg[b])
    eager = m16.generate(**batch(torch.bfloat16), max_new_tokens=n, return_logits=True, forced_tokens=r32.sequences,
                         use_graph=False)
    assert torch.equal(eager.logits, r16.logits) and torch.equal(eager.sequences, r16.sequences)


def _tiny_8b_like():
    """GAR-8B's structure at tiny sizes: PE-G style ViT (head_dim 96, NO cls token), Llama-3.1-8B style text model
    (head_dim 128, GQA 4:1, untied lm_head)."""
    from gar_amd import GARConfig
    return GARConfig.tiny(**{"vision.embed_dim": 192, "vision.num_heads": 2, "vision.mlp_dim": 448,
                             "vision_use_cls_token": False,
                             "text.hidden_size": 256, "text.num_attention_heads": 4, "text.num_key_value_heads": 1,
                             "text.head_dim": 128, "text.intermediate_size": 448, "text.tie_word_embeddings": False})


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
def test_gar8b_structure_tiny(dtype):
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = _tiny_8b_like()
    assert cfg.mllm_config.vision_config.head_dim == 96 and not cfg.mllm_config.vision_use_cls_token
    W = synthetic_weights(cfg)
    assert "mllm.lm_head.weight" in W
    proc = GARProcessor.from_config(cfg, max_num_tiles=4)
    s = _sample(cfg, proc, TINY8B_SINGLE, dtype=dtype)
    Wq = {k: v.to(dtype).float() for k, v in W.items()}
    ref_seq, ref_logits = _oracle(Wq, cfg, s, 12 if dtype == torch.float32 else 8)
    m = GARModel(cfg, W, dtype)
    out = m.generate(**s, max_new_tokens=ref_seq.shape[1], return_logits=True)
    if dtype == torch.float32:
        assert_discriminating(ref_seq, ref_logits)
        _check_f32(out, ref_seq, ref_logits, "graph")
        _check_f32(m.generate(**s, max_new_tokens=12, return_logits=True, use_graph=False), ref_seq, ref_logits, "eager")
    else:
        assert _rel_l2(out.logits.cpu()[:, 0], ref_logits[:, 0]) < (FP16_LOGIT_TOL if dtype == torch.float16 else BF16_LOGIT_TOL)
    g = m.generate(**s, max_new_tokens=ref_seq.shape[1])
    assert torch.equal(g.sequences.cpu(), out.sequences.cpu())


def test_bf16_gar8b_dims_one_layer():
    """GAR-8B shapes (BASELINE.json configs[3]: PE-G/14 ViT 1536 wide / 16 heads x 96 / MLP 8960 / no cls token,
    Llama-3.1-8B 4096 wide / 32 q + 8 kv heads x 128 / FFN 14336 / untied head; max_num_tiles=8 -> 5 tiles) with one
    layer each, bf16 against the f32 oracle on the bf16-rounded weights."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_8b(**{"vision.depth": 1, "text.num_hidden_layers": 1})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=8)
    s = _sample(cfg, proc, 0, 1024, 1024, dtype=torch.bfloat16)
    assert s["pixel_values"].shape[0] == 5
    Wq = {k: v.to(torch.bfloat16).float() for k, v in W.items()}
    ref_seq, ref_logits = _oracle(Wq, cfg, s, 2, attn_impl="sdpa")
    m = GARModel(cfg, W, torch.bfloat16)
    out = m.generate(**s, max_new_tokens=2, return_logits=True)
    assert _rel_l2(out.logits.cpu()[:, 0], ref_logits[:, 0]) < BF16_LOGIT_TOL
    g = m.generate(**s, max_new_tokens=2)
    assert torch.equal(g.sequences.cpu(), out.sequences.cpu())


def test_bf16_folded_norms_gar1b_dims():
    """The LayerNorms of the ViT blocks and the RMSNorms of the Llama prefill folded into the GEMMs around them
    (GARModel.FOLD_NORMS: consumer GEMMs read the residual stream with gamma-carrying, row-centred weights and scale their
    accumulator rows by rstd; producer GEMMs write the row statistics) against the same model with stand-alone norm passes
    and against the f32 oracle on the bf16-rounded weights — GAR-1B dims, two ViT blocks and two Llama layers so that the
    statistics of one block's proj / fc2 (o / down) feed the next block's qkv / fc1 (qkv / gate-up)."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 2, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    s = _sample(cfg, proc, 3, 1024, 1024, dtype=torch.bfloat16)
    Wq = {k: v.to(torch.bfloat16).float() for k, v in W.items()}
    ref_seq, ref_logits = _oracle(Wq, cfg, s, 1, attn_impl="sdpa")
    m = GARModel(cfg, W, torch.bfloat16, keep_plain_weights=True)      # the un-folded copies: only kept for this A/B
    assert m.FOLD_NORMS and "qkv_wf" in m.vblocks[0] and "gu_f" in m.layers[0] and "qkv_w" in m.vblocks[0] and "gu" in m.layers[0]
    folded = m.generate(**s, max_new_tokens=4, return_logits=True)
    assert "rstd" in m._ws[("vit",)] and "rstd" in m._ws[("prefill",)] and "h" not in m._ws[("vit",)]      # the folded path ran
    feats_f = m.get_image_features(s["pixel_values"], s["global_mask_values"]).clone()
    m.FOLD_NORMS = False
    plain = m.generate(**s, max_new_tokens=4, return_logits=True)
    feats_p = m.get_image_features(s["pixel_values"], s["global_mask_values"]).clone()
    # both are bf16 evaluations of the same function with different rounding points
    ef, ep = _rel_l2(folded.logits.cpu()[:, 0], ref_logits[:, 0]), _rel_l2(plain.logits.cpu()[:, 0], ref_logits[:, 0])
    print(f"folded norms: features folded vs plain rel-L2 {_rel_l2(feats_f, feats_p):.3e}; first-token logits vs the f32 oracle: "
          f"folded {ef:.3e}, plain {ep:.3e}; folded vs plain {_rel_l2(folded.logits[:, 0], plain.logits[:, 0]):.3e}")
    assert _rel_l2(feats_f, feats_p) < 1.5e-2
    assert ef < BF16_LOGIT_TOL and ep < BF16_LOGIT_TOL
    assert _rel_l2(folded.logits[:, 0], plain.logits[:, 0]) < BF16_LOGIT_TOL          # two bf16 roundings of one function
    # a row's result does not depend on its neighbours or on the pass it rides in: two identical samples, bit-identical rows
    m.FOLD_NORMS = True
    two = dict(input_ids=torch.cat([s["input_ids"]] * 2), pixel_values=torch.cat([s["pixel_values"]] * 2),
               global_mask_values=torch.cat([s["global_mask_values"]] * 2), bboxes=s["bboxes"] * 2,
               aspect_ratios=torch.cat([s["aspect_ratios"]] * 2))
    o2 = m.generate(**two, max_new_tokens=2, return_logits=True)
    assert torch.equal(o2.logits[0], o2.logits[1]) and torch.equal(o2.logits[0, 0], folded.logits[0, 0])


def test_bf16_fused_llm_paths_gar1b_dims():
    """Round 3's fused Llama paths at GAR-1B dims (two layers each, 18 identical sequences so that the decode step takes the
    16 < B <= 64 schedule): the prefill qkv GEMM with RoPE + q scale + cache append in its epilogue (LLM_QKV_EPILOGUE), the
    decode attention that takes the raw qkv GEMM output (DECODE_ATTN_TAKES_QKV) and the post-attention RMSNorm inside the
    gate/up GEMM (DECODE_GU_NORM_FOLDED), each against the same model with that switch off. The attention fusion changes no
    arithmetic: bit-identical tokens and logits; the other two move rounding points: logits within the bf16 tolerance of each
    other and of the f32 oracle."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 2, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    s = _sample(cfg, proc, 3, 1024, 1024, dtype=torch.bfloat16)
    Wq = {k: v.to(torch.bfloat16).float() for k, v in W.items()}
    ref_seq, ref_logits = _oracle(Wq, cfg, s, 1, attn_impl="sdpa")
    B, n = 18, 4
    batch = dict(input_ids=torch.cat([s["input_ids"]] * B), pixel_values=torch.cat([s["pixel_values"]] * B),
                 global_mask_values=torch.cat([s["global_mask_values"]] * B), bboxes=s["bboxes"] * B,
                 aspect_ratios=torch.cat([s["aspect_ratios"]] * B))
    m = GARModel(cfg, W, torch.bfloat16, keep_plain_weights=True)      # DECODE_GU_NORM_FOLDED = False needs a stand-alone norm path
    assert m.LLM_QKV_EPILOGUE and m.DECODE_ATTN_TAKES_QKV and m.DECODE_GU_NORM_FOLDED and not m.qkv_f_permuted
    fused = m.generate(**batch, max_new_tokens=n, return_logits=True)
    assert "qkv" not in m._ws[("prefill",)]                       # the [B*S, (Hq + 2 Hkv) hd] intermediate was never allocated
    assert "down_partial" in m._ws[("decode", B)]
    lf = fused.logits.cpu()
    assert torch.isfinite(lf).all()
    for r in range(1, B):
        assert torch.equal(lf[r], lf[0]), r
    assert _rel_l2(lf[:1, 0], ref_logits[:, 0]) < BF16_LOGIT_TOL
    # teacher-forced on the fused run's tokens so that every step compares like with like
    m.DECODE_ATTN_TAKES_QKV = False
    m._graphs.clear()                                             # the captured decode step holds the old launch sequence
    two_launch = m.generate(**batch, max_new_tokens=n, return_logits=True, forced_tokens=fused.sequences)
    assert torch.equal(two_launch.logits.cpu(), lf) and torch.equal(two_launch.sequences.cpu(), fused.sequences.cpu())
    m.DECODE_ATTN_TAKES_QKV = True
    m.DECODE_GU_NORM_FOLDED = False
    m._graphs.clear()
    plain_norm = m.generate(**batch, max_new_tokens=n, return_logits=True, forced_tokens=fused.sequences)
    m.DECODE_GU_NORM_FOLDED = True
    m.LLM_QKV_EPILOGUE = False                                    # head_dim 64: qkv_f is in the natural order either way
    m._graphs.clear()
    two_kernel_prefill = m.generate(**batch, max_new_tokens=n, return_logits=True, forced_tokens=fused.sequences)
    assert "qkv" in m._ws[("prefill",)]
    m.LLM_QKV_EPILOGUE = True
    e_norm = max(_rel_l2(plain_norm.logits[:1, j], fused.logits[:1, j]) for j in range(n))
    e_qkv = max(_rel_l2(two_kernel_prefill.logits[:1, j], fused.logits[:1, j]) for j in range(n))
    print(f"fused Llama paths: norm-in-gate/up vs stand-alone norm rel-L2 {e_norm:.3e}; qkv epilogue vs gemm + qkv_post {e_qkv:.3e}")
    assert torch.equal(plain_norm.logits[:, 0].cpu(), lf[:, 0])   # the first token comes out of the prefill: same path
    assert 0.0 < e_norm < BF16_LOGIT_TOL and 0.0 < e_qkv < BF16_LOGIT_TOL      # > 0: the switched-off paths really ran


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_generation_pipeline_equals_sequential_generate(tiny, dt):
    """GenerationPipeline (decode loop of batch i on a second stream beside the prompt phase of batch i + 1, two KV-state
    slots): five batches of different content, sizes and prompt lengths, with and without EOS polling — tokens and logits
    bit-identical to one model.generate per batch, in order; afterwards plain generate still works on either slot."""
    from gar_amd.modeling_gar import GARModel, GenerationPipeline
    cfg, W, proc = tiny
    ss = [_sample(cfg, proc, i, dtype=dt) for i in (3, 4, 6, 7)]

    def batch(idx):
        return dict(input_ids=torch.cat([ss[i]["input_ids"] for i in idx]),
                    pixel_values=torch.cat([ss[i]["pixel_values"] for i in idx]),
                    global_mask_values=torch.cat([ss[i]["global_mask_values"] for i in idx]),
                    bboxes=[ss[i]["bboxes"][0] for i in idx], aspect_ratios=torch.cat([ss[i]["aspect_ratios"] for i in idx]))
    longer = batch([0, 2])                                   # a different prompt length: another (B, Smax) bucket only if it crosses 256
    longer["input_ids"] = torch.cat([longer["input_ids"], torch.randint(10, 290, (2, 5), generator=torch.Generator().manual_seed(11))], 1)
    batches = [batch([0, 1]), batch([2, 3]), batch([1, 0]), batch([3]), longer]
    m = GARModel(cfg, W, dt)
    for kw in (dict(max_new_tokens=6, return_logits=True),
               dict(max_new_tokens=6, return_logits=True, eos_token_id=[int(t) for t in range(0, cfg.mllm_config.text_config.vocab_size, 3)],
                    sync_every=2)):
        ref = [m.generate(**b, **kw) for b in batches]
        outs = m.generate_pipelined(batches, **kw)
        torch.cuda.synchronize()
        assert len(outs) == len(ref)
        for o, r in zip(outs, ref):
            assert torch.equal(o.sequences.cpu(), r.sequences.cpu()) and torch.equal(o.logits.cpu(), r.logits.cpu())
        # the incremental form bench.py uses
        pipe = GenerationPipeline(m, **kw)
        got = []
        for b in batches:
            got.extend(pipe.submit(b))
        got.extend(pipe.flush())
        torch.cuda.synchronize()
        for o, r in zip(got, ref):
            assert o.done is not None and torch.equal(o.sequences.cpu(), r.sequences.cpu())
    # either slot is an ordinary KV state
    assert torch.equal(m.generate(**batches[1], max_new_tokens=6).sequences.cpu(),
                       m.generate(**batches[1], max_new_tokens=6, state_slot=1).sequences.cpu())


def test_f32_parity_gar1b_dims_multi_region_one_layer():
    """BASELINE.json configs[2]: 4 masks per 1024^2 image, relationship prompt, GAR-1B shapes (one layer each):
    four 256-row RoI replays spliced into one ~5.5k-token sequence, f32 token parity with the oracle."""
    from gar_amd import GARConfig
    from gar_amd.eval_dataset import MultiRegionDataset
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.synthetic import RELATIONSHIP_QUESTION, synthetic_disjoint_masks, synthetic_image
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    s = MultiRegionDataset(synthetic_image(3), synthetic_disjoint_masks(3, 4), RELATIONSHIP_QUESTION, proc,
                           data_dtype=torch.float32, device="cpu")[0]
    assert s["pixel_values"].shape[0] == 17 and len(s["bboxes"][0]) == 4
    ref_seq, ref_logits = _oracle(W, cfg, s, 8, attn_impl="sdpa")
    assert_discriminating(ref_seq, ref_logits)
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True), ref_seq, ref_logits)


def test_f32_video_replay_gar8b_structure_tiny():
    """BASELINE.json configs[4] structure: video replay (8 frames, per-frame mask) on the GAR-8B-like tiny config."""
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    from oracle import gar_oracle as O
    cfg = _tiny_8b_like()
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=4)
    s = _video_sample(cfg, proc, 8, base=TINY8B_VIDEO_BASE)
    assert s["pixel_values"].shape[0] == 8 and len(s["bboxes"][0]) == 8
    ref_seq, ref_logits = O.gar_generate(W, cfg, s["pixel_values"], s["global_mask_values"], None, s["bboxes"],
                                         s["input_ids"], None, max_new_tokens=8, return_logits=True,
                                         video_frame_tokens=s["video_frame_tokens"])
    assert_discriminating(ref_seq, ref_logits)
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True), ref_seq, ref_logits)


def test_video_replay_gar8b_dims_one_layer():
    """BASELINE.json configs[4] at its real dimensions (one layer each): an 8-frame 1024^2 clip with a per-frame mask
    through the GAR-8B video replay — C = 4096, eight 256-row replays per clip (the 8-job instantiation of
    roi_replay_inplace at C = 4096), S ~ 4.3k, PE-G/14 head_dim 96 without cls token, Llama head_dim 128, untied head.
    (1) f32 HIP vs the oracle: tokens and per-step logits; (2) bf16 HIP vs the oracle on the bf16-rounded weights;
    (3) a batch of two clips (16 RoI jobs in one launch) equals the single runs."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    from oracle import gar_oracle as O
    cfg = GARConfig.gar_8b(**{"vision.depth": 1, "text.num_hidden_layers": 1})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=8)
    s = _video_sample(cfg, proc, 8, base=90, w=1024, h=1024)
    assert s["pixel_values"].shape[0] == 8 and len(s["bboxes"][0]) == 8 and s["input_ids"].shape[1] > 4200
    n = 4
    ref_seq, ref_logits = O.gar_generate(W, cfg, s["pixel_values"], s["global_mask_values"], None, s["bboxes"],
                                         s["input_ids"], None, max_new_tokens=n, return_logits=True,
                                         video_frame_tokens=s["video_frame_tokens"], attn_impl="sdpa")
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=n, return_logits=True), ref_seq, ref_logits, "f32 graph")
    # the replayed rows themselves, against the oracle's inputs_embeds
    proj = m.get_image_features(s["pixel_values"], s["global_mask_values"], pooled=False)
    emb = m.build_inputs_embeds(s["input_ids"], None, s["bboxes"], None, 8, True, s["video_frame_tokens"], proj=proj).cpu().clone()
    ref_emb = O.build_inputs_embeds(W, cfg, s["pixel_values"], s["global_mask_values"], None, s["bboxes"], s["input_ids"],
                                    video_frame_tokens=s["video_frame_tokens"])
    assert _rel_l2(emb, ref_emb) < 1e-5
    s2 = _video_sample(cfg, proc, 8, base=130, w=1024, h=1024)
    two = dict(input_ids=torch.cat([s["input_ids"], s2["input_ids"]]), pixel_values=torch.cat([s["pixel_values"], s2["pixel_values"]]),
               global_mask_values=torch.cat([s["global_mask_values"], s2["global_mask_values"]]),
               bboxes=s["bboxes"] + s2["bboxes"], feature_replay_video=True, video_frame_tokens=s["video_frame_tokens"])
    o2 = m.generate(**s2, max_new_tokens=n).sequences
    o12 = m.generate(**two, max_new_tokens=n).sequences
    assert o12[0].cpu().tolist() == ref_seq[0].tolist() and torch.equal(o12[1], o2[0])
    del m
    torch.cuda.empty_cache()
    Wq = {k: v.to(torch.bfloat16).float() for k, v in W.items()}
    refq_seq, refq_logits = O.gar_generate(Wq, cfg, s["pixel_values"].to(torch.bfloat16).float(),
                                           s["global_mask_values"].to(torch.bfloat16).float(), None, s["bboxes"],
                                           s["input_ids"], None, max_new_tokens=1, return_logits=True,
                                           video_frame_tokens=s["video_frame_tokens"], attn_impl="sdpa")
    sb = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in s.items()}
    mb = GARModel(cfg, W, torch.bfloat16)
    ob = mb.generate(**sb, max_new_tokens=n, return_logits=True)
    assert _rel_l2(ob.logits.cpu()[:, 0], refq_logits[:, 0]) < BF16_LOGIT_TOL
    assert torch.equal(mb.generate(**sb, max_new_tokens=n, use_graph=False).sequences, ob.sequences)


def test_replica_from_shapes_after_weight_copy(tiny):
    """the non-source ranks of the data-parallel runner build the model from shapes only and receive the PREPARED weight
    tensors by RCCL broadcast (bench.py): emulate the broadcast with copies and require identical captions — also for
    the GAR-8B-like structure, whose prepared tensors are padded / untied."""
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    for cfg in (tiny[0], _tiny_8b_like()):
        W = synthetic_weights(cfg)
        proc = GARProcessor.from_config(cfg, max_num_tiles=4)
        s = _sample(cfg, proc, 4, dtype=torch.bfloat16)
        src = GARModel(cfg, W, torch.bfloat16)
        dst = GARModel.from_shapes(cfg, torch.bfloat16)
        a, b = src.weight_tensors(), dst.weight_tensors()
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert x.shape == y.shape and x.dtype == y.dtype
            y.copy_(x)
        o0 = src.generate(**s, max_new_tokens=6, return_logits=True)
        o1 = dst.generate(**s, max_new_tokens=6, return_logits=True)
        assert torch.equal(o0.sequences, o1.sequences) and torch.equal(o0.logits, o1.logits)


@pytest.mark.parametrize("name", ["gar_1b", "gar_8b"])
def test_bf16_device_weights_are_one_copy(name):
    """bf16 keeps ONE copy of every weight (the norm-folded forms serve the tile GEMMs, the decode GEMVs and the small-shape
    fallback): the prepared tensors a replica holds — and the RCCL broadcast moves — are within 3 % of the checkpoint's bytes
    (round 3 kept plain + folded copies: +1.6 GB GAR-1B, +11 GB GAR-8B)."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.weights import weight_shapes
    cfg = getattr(GARConfig, name)()
    ckpt = sum(int(torch.tensor(s).prod()) for s in weight_shapes(cfg).values()) * 2
    m = GARModel.from_shapes(cfg, torch.bfloat16)
    held = sum(t.numel() * t.element_size() for t in m.weight_tensors())
    print(f"{name}: checkpoint {ckpt / 2**30:.2f} GiB, prepared device weights {held / 2**30:.2f} GiB ({held / ckpt - 1:+.2%})")
    assert "qkv_w" not in m.vblocks[0] and "gu" not in m.layers[0] and "qkv" not in m.layers[0]
    assert abs(held / ckpt - 1) < 0.03
    del m
    torch.cuda.empty_cache()


def test_f32_parity_gar1b_dims_max_tiles_one_layer():
    """PLM's default max_num_tiles=36 (6x6 canvas -> 37 tiles, S ~ 9.8k) at GAR-1B shapes with one
    layer each: the largest single-region configuration, f32 token parity with the oracle."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=36)
    s = _sample(cfg, proc, 2, 1024, 1024)
    assert s["pixel_values"].shape[0] == 37 and s["input_ids"].shape[1] > 9700
    ref_seq, ref_logits = _oracle(W, cfg, s, 4, attn_impl="sdpa")
    assert_discriminating(ref_seq, ref_logits)
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=4, return_logits=True), ref_seq, ref_logits)


def test_edge_cases_tiny(tiny):
    """single new token (no decode loop), text-only prompt (no pixel_values: LlamaModel over token embeddings only),
    and a batch of identical samples in bf16 whose rows must be identical."""
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    s = _sample(cfg, proc, 3)
    m = GARModel(cfg, W, torch.float32)
    ref_seq, _ = _oracle(W, cfg, s, 1)
    assert m.generate(**s, max_new_tokens=1).sequences.cpu().tolist() == ref_seq.tolist()
    ids = torch.tensor([[296, 306, 20, 21, 22, 307, 101, 102, 103, 309]], dtype=torch.int64)
    tcfg = cfg.mllm_config.text_config
    cache = O.KVCache(tcfg.num_hidden_layers)
    emb = torch.nn.functional.embedding(ids, W[O.LM + "embed_tokens.weight"])
    toks = []
    h = O.llama_forward(emb, W, tcfg, cache, "eager")
    head = O.lm_head_weight(W, tcfg)
    for _ in range(5):
        nxt = torch.argmax(torch.nn.functional.linear(h[:, -1], head), -1)
        toks.append(int(nxt))
        h = O.llama_forward(torch.nn.functional.embedding(nxt, W[O.LM + "embed_tokens.weight"]).unsqueeze(1), W, tcfg,
                            cache, "eager")
    out = m.generate(input_ids=ids, max_new_tokens=5)
    assert out.sequences.cpu().tolist() == [toks]
    mb = GARModel(cfg, W, torch.bfloat16)
    sb = _sample(cfg, proc, 3, dtype=torch.bfloat16)
    three = dict(input_ids=torch.cat([sb["input_ids"]] * 3), pixel_values=torch.cat([sb["pixel_values"]] * 3),
                 global_mask_values=torch.cat([sb["global_mask_values"]] * 3), bboxes=sb["bboxes"] * 3,
                 aspect_ratios=torch.cat([sb["aspect_ratios"]] * 3))
    seq = mb.generate(**three, max_new_tokens=6).sequences
    assert torch.equal(seq[0], seq[1]) and torch.equal(seq[0], seq[2])


@pytest.mark.parametrize("vision_bias", ["split", "unfused"])
def test_from_pretrained_hf_style_sharded_checkpoint(tiny, tmp_path, vision_bias, hf_style_checkpoint):
    """A checkpoint directory the way a release is laid out — config.json whose vision model_args only hold the
    overrides the reference reads, safetensors shards, timm-Eva bias variants, unused tensors — loads through
    GARModel.from_pretrained / GARProcessor.from_pretrained and generates the oracle's tokens."""
    import json
    from safetensors.torch import save_file
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    cfg, W, proc = tiny
    ck, d = hf_style_checkpoint(cfg, W, vision_bias)
    keys = sorted(ck)
    for i in range(3):                                           # three shards, keys interleaved
        save_file({k: ck[k].contiguous() for k in keys[i::3]}, str(tmp_path / f"model-{i + 1:05d}-of-00003.safetensors"))
    (tmp_path / "config.json").write_text(json.dumps(d))
    m = GARModel.from_pretrained(str(tmp_path), torch.float32)
    assert m.config.mllm_config.vision_config.depth == 2
    p2 = GARProcessor.from_pretrained(str(tmp_path), m.config, max_num_tiles=4)
    s = _sample(cfg, proc, 2)
    s2 = _sample(m.config, p2, 2)
    assert all(torch.equal(s[k], s2[k]) for k in ("pixel_values", "global_mask_values", "input_ids"))
    assert s["bboxes"] == s2["bboxes"]
    ref_seq, _ = _oracle(W, cfg, s, 8)
    out = m.generate(**s2, max_new_tokens=8)
    assert out.sequences.cpu().tolist() == ref_seq.tolist()


# tolerances at full depth (23 ViT + 16 Llama layers): f32 rounding differences accumulate through 39 layers, stated
# separately from the per-layer F32_LOGIT_TOL; measured values are printed by the test (pytest -s) and quoted in DESIGN.md
FULL_DEPTH_F32_TOL = 2e-4     # measured 7.8e-6 over 16 teacher-forced tokens
FULL_DEPTH_BF16_REL_L2 = 5.3e-2   # measured: first token 3.2e-2 .. 3.9e-2, worst of 8 x 64 steps 4.4e-2 (B = 64: 4.5e-2; one region in round 5: 4.8e-2)
# Top-1 agreement floors (DESIGN.md section 7 holds the table: statistic, n, measured value, sigma, floor, the commit that measured it).
# A floor is the measured value minus THREE standard deviations of a binomial with the test's own n — wide enough that builds which
# differ only in an fp32 rounding order pass, narrow enough that a kernel regression worth 5 points fails — and is FROZEN: a build
# that falls below it has to be explained, the floor does not follow it.
FULL_DEPTH_BF16_AGREE_B1 = 0.87        # B = 1, 8 regions x 64 steps (n = 512): measured 0.9082 (465 of 512), sigma 0.0128
FULL_DEPTH_BF16_AGREE = 0.929          # B = 64, 64 regions x 64 steps (n = 4096): measured 0.9404 (rounds 4, 5 and 6 alike), sigma 0.0037
FULL_DEPTH_BF16_AGREE_WORST_REGION = 0.80      # the MINIMUM of 64 per-region statistics of 64 steps each (sigma 0.03 per region): measured 0.859, 0.844
FULL_DEPTH_ORACLE_TOKENS = 16          # tokens of the full-depth f32 run that meet the CPU oracle (rounds 3 - 5: 4)
FULL_DEPTH_B1_REGIONS = 8


def test_full_depth_f32_vs_oracle_and_bf16_vs_f32_teacher_forced():
    """Full GAR-1B (23 + 16 layers, 17 tiles, S ~ 4.7k), the configuration bench.py times, one region per call (the reference's
    calling pattern, demo/gar_with_mask.py:112-122):
    (1) f32 HIP vs the CPU oracle over 16 tokens: the HIP run is fed the ORACLE's tokens, so every one of the 16 steps is compared on
        the same context — logits within FULL_DEPTH_F32_TOL, the same token wherever the oracle's top-2 margin is not a near-tie —
        and the free-running f32 caption starts with the oracle's tokens;
    (2) bf16 HIP vs f32 HIP over whole 64-token captions of 8 DISTINCT regions on identical contexts (the bf16 model is fed the f32
        model's tokens): n = 512 steps of top-1 agreement, and every disagreement must sit at an f32 top-2 margin below twice that
        region's measured bf16 logit error."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    from parity_util import F32_LOGIT_TOL as TOL, MARGIN_FACTOR, discrimination_stats
    cfg = GARConfig.gar_1b()
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    R, NO = FULL_DEPTH_B1_REGIONS, FULL_DEPTH_ORACLE_TOKENS
    ss = [_sample(cfg, proc, i, 1024, 1024) for i in range(R)]
    s = ss[0]
    assert s["pixel_values"].shape[0] == 17
    ref_seq, ref_logits = _oracle(W, cfg, s, NO, attn_impl="sdpa")
    assert_discriminating(ref_seq[:, :4], ref_logits[:, :4])
    m32 = GARModel(cfg, W, torch.float32)
    tf = m32.generate(**s, max_new_tokens=NO, return_logits=True, forced_tokens=ref_seq)       # every step on the oracle's context
    err = float((tf.logits.cpu() - ref_logits).abs().max()) / float(ref_logits.abs().max())
    top2 = ref_logits.topk(2, -1).values[0]
    clear = (top2[:, 0] - top2[:, 1]) >= MARGIN_FACTOR * TOL * float(ref_logits.abs().max())     # steps that are not near-ties
    print(f"full depth f32 HIP vs oracle over {NO} teacher-forced tokens: max|dlogit| / max|logit| = {err:.3e}; "
          f"{int(clear.sum())} of {NO} steps have a top-2 margin above {MARGIN_FACTOR} x the tolerance")
    assert err <= FULL_DEPTH_F32_TOL, err
    assert int(clear.sum()) >= NO - 2
    assert torch.equal(tf.sequences.cpu()[0][clear], ref_seq[0][clear])
    seq32s, lg32s = [], []
    for i, si in enumerate(ss):
        o32 = m32.generate(**si, max_new_tokens=64, return_logits=True)
        seq32s.append(o32.sequences.clone())
        lg32s.append(o32.logits.cpu())
        if i == 0:      # free-running: the oracle's tokens up to its first near-tie
            k = NO if bool(clear.all()) else int((~clear).nonzero()[0])
            assert o32.sequences[0, :k].cpu().tolist() == ref_seq[0, :k].tolist()
            distinct, repeats, rel_margin = discrimination_stats(seq32s[0][0].tolist(), lg32s[0][0])
            print(f"full depth f32 caption: {distinct} distinct of 64 tokens, {repeats} immediate repeats, min top-2 margin "
                  f"{rel_margin:.2e} x max|logit|")
            assert repeats == 0 and distinct >= 52                   # no fixed point over the whole caption
    assert len({tuple(q[0].tolist()) for q in seq32s}) == R          # 8 different regions -> 8 different captions
    del m32
    torch.cuda.empty_cache()
    m16 = GARModel(cfg, W, torch.bfloat16)
    n_agree, worst_rel, first_rel = 0, 0.0, []
    for i, si in enumerate(ss):
        sb = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in si.items()}
        o16 = m16.generate(**sb, max_new_tokens=64, return_logits=True, forced_tokens=seq32s[i])
        lg16, lg32 = o16.logits.cpu(), lg32s[i]
        rel = [_rel_l2(lg16[:, j], lg32[:, j]) for j in range(64)]
        agree = (o16.sequences == seq32s[i])[0].cpu()
        max_err = float((lg16 - lg32).abs().max())
        t2 = lg32.topk(2, -1).values[0]
        margins = t2[:, 0] - t2[:, 1]
        for j in (~agree).nonzero().flatten().tolist():
            assert float(margins[j]) < 2 * max_err, (i, j, float(margins[j]), max_err)
        n_agree += int(agree.sum())
        worst_rel = max(worst_rel, max(rel))
        first_rel.append(rel[0])
    rate = n_agree / (64.0 * R)
    print(f"full depth bf16 vs f32 (teacher forced, {R} regions x 64 tokens, B = 1): top-1 agreement {rate:.4f} ({n_agree} of {64 * R}), "
          f"first-token rel-L2 {min(first_rel):.3e} .. {max(first_rel):.3e}, worst rel-L2 {worst_rel:.3e}")
    assert worst_rel < FULL_DEPTH_BF16_REL_L2, worst_rel
    assert rate >= FULL_DEPTH_BF16_AGREE_B1, rate          # a regression in a bf16 kernel must not hide below the measured value
    # free-running bf16 through the graph == the same model run eagerly (bit-identical kernels)
    sb = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in ss[0].items()}
    free_g = m16.generate(**sb, max_new_tokens=16)
    free_e = m16.generate(**sb, max_new_tokens=16, use_graph=False)
    assert torch.equal(free_g.sequences, free_e.sequences)


def test_bench_configuration_bf16_vs_f32_teacher_forced_64_regions():
    """Parity AT THE CONFIGURATION bench.py TIMES: full-depth GAR-1B, B = 64 DISTINCT synthetic regions in
    one generate — the planner's default passes (387 / 383 / 318 image tiles, 26 / 26 / 12 sequences), the 16 < B <= 64 decode
    schedule (norm-folded GEMVs, split-K `down` + reduce, single-split decode attention at S ~ 4.7k) replayed from the hipGraph,
    the pruned last prefill layer — bf16 teacher-forced on the f32 HIP run of the SAME batch (itself within 1e-5 of the CPU oracle
    at B = 1, test_full_depth_...): per-row, per-step relative L2 of the logits, top-1 agreement over all 64 x 64 steps, and
    every disagreement at an f32 top-2 margin inside the measured bf16 logit error — the B = 1 test's assertions, at B = 64."""
    from gar_amd import GARConfig
    from gar_amd.eval_dataset import SingleRegionCaptionDataset
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b()
    W = synthetic_weights(cfg)
    B, NT = 64, 64
    proc = GARProcessor.from_config(cfg, max_num_tiles=16).use_gpu_preprocessing("cuda:0", torch.float32)
    ss = [SingleRegionCaptionDataset(synthetic_image(700 + i), synthetic_mask(700 + i), proc, data_dtype=torch.float32,
                                     device="cuda:0")[0] for i in range(B)]
    batch = dict(input_ids=torch.cat([x["input_ids"] for x in ss]), pixel_values=torch.cat([x["pixel_values"] for x in ss]),
                 global_mask_values=torch.cat([x["global_mask_values"] for x in ss]), bboxes=[x["bboxes"][0] for x in ss],
                 aspect_ratios=torch.cat([x["aspect_ratios"] for x in ss]))
    del ss
    assert batch["pixel_values"].shape[0] == 17 * B
    m32 = GARModel(cfg, W, torch.float32)
    o32 = m32.generate(**batch, max_new_tokens=NT, return_logits=True)
    seq32 = o32.sequences.clone()
    lg32 = o32.logits                                        # [B, NT, V] fp32 on the device (2.1 GB)
    assert len({tuple(r) for r in seq32.cpu().tolist()}) == B           # 64 different regions -> 64 different captions
    del m32, o32
    torch.cuda.empty_cache()
    bb = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    m16 = GARModel(cfg, W, torch.bfloat16)
    S = batch["input_ids"].shape[1]
    plan_v, plan_l = m16._plan_passes(B, 17, S)
    assert len(plan_v) > 1 and len(plan_l) > 1 and sum(plan_v) == 17 * B and sum(plan_l) == B      # the planner's passes, not one chunk
    o16 = m16.generate(**bb, max_new_tokens=NT, return_logits=True, forced_tokens=seq32)
    assert "down_partial" in m16._ws[("decode", B)] and m16._graphs                  # the B = 64 schedule, replayed from the graph
    lg16 = o16.logits
    rel = ((lg16 - lg32).double().norm(dim=-1) / lg32.double().norm(dim=-1))         # [B, NT]
    agree = (o16.sequences == seq32)
    rate = float(agree.float().mean())
    err_row = (lg16 - lg32).abs().amax(dim=(1, 2))                                   # [B] max |dlogit| of each region's caption
    top2 = lg32.topk(2, -1).values
    margins = top2[..., 0] - top2[..., 1]                                            # [B, NT]
    print(f"B = 64 full depth bf16 vs f32 (teacher forced, {NT} tokens x {B} regions): top-1 agreement {rate:.4f} (worst region "
          f"{float(agree.float().mean(1).min()):.3f}), rel-L2 first token mean {float(rel[:, 0].mean()):.3e} / worst {float(rel[:, 0].max()):.3e}, "
          f"worst of all steps {float(rel.max()):.3e}, max|dlogit| {float(err_row.max()):.3e}")
    assert float(rel.max()) < FULL_DEPTH_BF16_REL_L2, float(rel.max())
    bad = (~agree) & (margins >= 2 * err_row[:, None])
    assert not bool(bad.any()), bad.nonzero().tolist()[:8]
    assert rate >= FULL_DEPTH_BF16_AGREE, rate
    worst_region = float(agree.float().mean(1).min())
    assert worst_region >= FULL_DEPTH_BF16_AGREE_WORST_REGION, worst_region
    # the same batch through eager launches: the graph replays exactly these kernels
    e16 = m16.generate(**bb, max_new_tokens=8, use_graph=False, forced_tokens=seq32)
    assert torch.equal(e16.sequences, o16.sequences[:, :8])


@pytest.mark.parametrize("max_num_tiles,canvas", [(16, (4, 4)), (8, (3, 2))])
def test_config0_demo_asset_f32_parity(golden_dir, max_num_tiles, canvas):
    """BASELINE configs[0]: assets/demo_image_1.png (1024 x 770 RGBA) + demo_mask_1.png through the sample builder the
    demo CLI uses (demo/gar_with_mask.py:74-128 of the reference), GAR-1B dims (one ViT + two Llama layers): HIP f32 vs
    the oracle on the (4, 4) canvas of the release config and on the non-square (3, 2) canvas of max_num_tiles=8."""
    import os
    import numpy as np
    from PIL import Image
    from gar_amd import GARConfig
    from gar_amd.eval_dataset import SingleRegionCaptionDataset
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 2})
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=max_num_tiles)
    img = Image.open(os.path.join(golden_dir, "demo_image_1.png"))
    mask = np.array(Image.open(os.path.join(golden_dir, "demo_mask_1.png")).convert("L")).astype(bool)
    s = SingleRegionCaptionDataset(img, mask, proc, data_dtype=torch.float32, device="cpu")[0]
    assert s["aspect_ratios"].tolist() == [list(canvas)]
    assert s["bboxes"][0]["128005"] == (0.720703125, 0.8688311688311688, 0.7939453125, 0.9233766233766234)
    ref_seq, ref_logits = _oracle(W, cfg, s, 8, attn_impl="sdpa")
    m = GARModel(cfg, W, torch.float32)
    _check_f32(m.generate(**s, max_new_tokens=8, return_logits=True), ref_seq, ref_logits, "host preprocessing")
    # the device preprocessing path (raw uint8 image -> tiles on the GPU) feeds the same tokens
    proc.use_gpu_preprocessing("cuda:0", torch.float32)
    sd = SingleRegionCaptionDataset(img, mask, proc, data_dtype=torch.float32, device="cuda:0")[0]
    assert torch.equal(sd["pixel_values"].cpu(), s["pixel_values"])
    assert m.generate(**sd, max_new_tokens=8).sequences.cpu().tolist() == ref_seq.tolist()


FULL_DEPTH_8B_F32_TOL = 2e-4
# 47 + 32 layers of bf16 rounding (GAR-1B's 23 + 16: bound 5.3e-2, measured 4.8e-2). Measured here: first token 7.1e-2 ... 7.2e-2,
# worst step 7.7e-2 ... 8.0e-2 across builds that differ only in an fp32 summation order: bound = the worst measured value + 10 %
FULL_DEPTH_8B_BF16_REL_L2 = 8.8e-2
# 16 DISTINCT regions x 32 teacher-forced steps in one batch (n = 512; rounds 3 - 5 ran 32 steps of ONE sequence, sigma 0.055, and the
# floor followed the builds down to 0.78). Measured 0.8359 (428 of 512; sigma 0.0164; the worst of the 16 regions 0.688 = 22 of 32): the
# population value sits BELOW what the single sequence of rounds 3 - 5 happened to show (0.844 ... 0.969) — 79 layers of bf16 rounding
# (rel-L2 of the logits 5.3e-2 ... 7.7e-2) against the synthetic model's top-2 margins. Floor = measured - 3 sigma, frozen (DESIGN.md
# section 7); what a disagreeing step may look like is pinned separately (its f32 margin < 2 x its region's largest logit error).
FULL_DEPTH_8B_REGIONS = 16
FULL_DEPTH_8B_BF16_AGREE = 0.787


def test_full_depth_gar8b_f32_vs_oracle_and_bf16_vs_f32_teacher_forced():
    """Full GAR-8B (BASELINE.json configs[3]: 47 PE-G/14 layers with head_dim 96 and no cls token + 32 Llama-3.1-8B layers
    with head_dim 128 and an untied head; max_num_tiles=8 -> 5 tiles, S ~ 1.6k):
    (1) f32 HIP vs the CPU oracle — the first 2 greedy tokens (prefill + one decode step) and their logits;
    (2) bf16 HIP vs f32 HIP teacher-forced over 32 tokens of 16 DISTINCT regions in one batch (n = 512 steps): per-step relative L2
        of the logits, top-1 agreement, every disagreement at an f32 margin inside that region's measured bf16 logit error."""
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    cfg = GARConfig.gar_8b()
    W = synthetic_weights(cfg)
    proc = GARProcessor.from_config(cfg, max_num_tiles=8)
    R = FULL_DEPTH_8B_REGIONS
    ss = [_sample(cfg, proc, i, 1024, 1024) for i in range(R)]
    s = ss[0]
    assert s["pixel_values"].shape[0] == 5
    ref_seq, ref_logits = _oracle(W, cfg, s, 2, attn_impl="sdpa")
    NT = 32
    m32 = GARModel(cfg, W, torch.float32)
    o1 = m32.generate(**s, max_new_tokens=2, return_logits=True)
    err = float((o1.logits.cpu() - ref_logits).abs().max()) / float(ref_logits.abs().max())
    print(f"GAR-8B full depth f32 HIP vs oracle: max|dlogit| / max|logit| = {err:.3e}")
    assert o1.sequences.cpu().tolist() == ref_seq.tolist()
    assert err <= FULL_DEPTH_8B_F32_TOL, err
    batch = dict(input_ids=torch.cat([x["input_ids"] for x in ss]), pixel_values=torch.cat([x["pixel_values"] for x in ss]),
                 global_mask_values=torch.cat([x["global_mask_values"] for x in ss]), bboxes=[x["bboxes"][0] for x in ss],
                 aspect_ratios=torch.cat([x["aspect_ratios"] for x in ss]))
    o32 = m32.generate(**batch, max_new_tokens=NT, return_logits=True)
    seq32, lg32 = o32.sequences.clone(), o32.logits.cpu()                 # [R, NT], [R, NT, V]
    assert seq32[0, :2].cpu().tolist() == ref_seq[0].tolist()             # the batched row 0 = the single run = the oracle
    assert len({tuple(r) for r in seq32.cpu().tolist()}) == R
    from parity_util import discrimination_stats
    distinct, repeats, rel_margin = discrimination_stats(seq32[0].tolist(), lg32[0])
    print(f"GAR-8B full depth f32 caption: {distinct} distinct of {NT} tokens, {repeats} immediate repeats, min top-2 "
          f"margin {rel_margin:.2e} x max|logit|")
    del m32, o32, o1
    torch.cuda.empty_cache()
    bb = {k: (v.to(torch.bfloat16) if torch.is_tensor(v) and v.is_floating_point() else v) for k, v in batch.items()}
    m16 = GARModel(cfg, W, torch.bfloat16)
    del W
    o16 = m16.generate(**bb, max_new_tokens=NT, return_logits=True, forced_tokens=seq32)
    lg16 = o16.logits.cpu()
    rel = ((lg16 - lg32).double().norm(dim=-1) / lg32.double().norm(dim=-1))         # [R, NT]
    agree = (o16.sequences == seq32).cpu()
    rate = float(agree.float().mean())
    err_row = (lg16 - lg32).abs().amax(dim=(1, 2))                                   # [R]
    top2 = lg32.topk(2, -1).values
    margins = top2[..., 0] - top2[..., 1]
    print(f"GAR-8B full depth bf16 vs f32 (teacher forced, {R} regions x {NT} tokens): top-1 agreement {rate:.4f} ({int(agree.sum())} of "
          f"{R * NT}; worst region {float(agree.float().mean(1).min()):.3f}), first-token rel-L2 {float(rel[:, 0].min()):.3e} .. "
          f"{float(rel[:, 0].max()):.3e}, worst rel-L2 {float(rel.max()):.3e}, max|dlogit| {float(err_row.max()):.3e}")
    assert float(rel.max()) < FULL_DEPTH_8B_BF16_REL_L2, float(rel.max())
    bad = (~agree) & (margins >= 2 * err_row[:, None])
    assert not bool(bad.any()), bad.nonzero().tolist()[:8]
    assert rate >= FULL_DEPTH_8B_BF16_AGREE, rate          # a regression in the head_dim 96 / 128 kernels must not hide below it
    sb = {k: (v[:1] if torch.is_tensor(v) and k != "pixel_values" and k != "global_mask_values" else v) for k, v in bb.items()}
    sb["pixel_values"], sb["global_mask_values"], sb["bboxes"] = bb["pixel_values"][:5], bb["global_mask_values"][:5], bb["bboxes"][:1]
    free_g = m16.generate(**sb, max_new_tokens=8)
    free_e = m16.generate(**sb, max_new_tokens=8, use_graph=False)
    assert torch.equal(free_g.sequences, free_e.sequences)


def test_bf16_decode_above_64_sequences(tiny):
    """more than 64 sequences per step exceed the split-K `down` schedule (M <= 64 rows) — generate must
    fall back to gemm(EPI_RES) + rmsnorm instead of raising on the first decode step. 65 rows in bf16, teacher-forced on
    the f32 HIP run of the same batch: per-step logits within the bf16 tolerance, rows of the same sample identical."""
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    ss = [_sample(cfg, proc, i) for i in (3, 4, 6, 7)]
    order = [i % 4 for i in range(65)]

    def batch(dt):
        return dict(input_ids=torch.cat([ss[i]["input_ids"] for i in order]),
                    pixel_values=torch.cat([ss[i]["pixel_values"] for i in order]).to(dt),
                    global_mask_values=torch.cat([ss[i]["global_mask_values"] for i in order]).to(dt),
                    bboxes=[ss[i]["bboxes"][0] for i in order],
                    aspect_ratios=torch.cat([ss[i]["aspect_ratios"] for i in order]))
    n = 5
    r32 = GARModel(cfg, W, torch.float32).generate(**batch(torch.float32), max_new_tokens=n, return_logits=True)
    m = GARModel(cfg, W, torch.bfloat16)
    out = m.generate(**batch(torch.bfloat16), max_new_tokens=n, return_logits=True, forced_tokens=r32.sequences)
    assert "down_partial" not in m._ws[("decode", 65)]
    lg = out.logits.cpu()
    assert torch.isfinite(lg).all()
    for j in range(n):
        assert _rel_l2(out.logits[:, j], r32.logits[:, j]) < TINY_BF16_STEP_REL_L2, j
    for r in range(4, 65):
        assert torch.equal(lg[r], lg[r % 4]), r


def _lengthen(s, extra, seed):
    """the sample with `extra` plain text tokens appended to its prompt (different prompt lengths for one image size)"""
    if extra == 0:
        return dict(s)
    g = torch.Generator().manual_seed(seed)
    tail = torch.randint(10, 290, (1, extra), generator=g, dtype=torch.int64)
    out = dict(s)
    out["input_ids"] = torch.cat([s["input_ids"], tail], dim=1)
    if out.get("attention_mask") is not None:
        out["attention_mask"] = torch.ones_like(out["input_ids"])
    return out


def _left_pad_batch(samples, pad_id=7):
    S = max(x["input_ids"].shape[1] for x in samples)
    ids = torch.full((len(samples), S), pad_id, dtype=torch.int64)
    mask = torch.zeros(len(samples), S, dtype=torch.int64)
    for b, x in enumerate(samples):
        n = x["input_ids"].shape[1]
        ids[b, S - n:] = x["input_ids"][0]
        mask[b, S - n:] = 1
    return dict(input_ids=ids, attention_mask=mask, pixel_values=torch.cat([x["pixel_values"] for x in samples]),
                global_mask_values=torch.cat([x["global_mask_values"] for x in samples]),
                bboxes=sum((x["bboxes"] for x in samples), []),
                aspect_ratios=torch.cat([x["aspect_ratios"] for x in samples]))


def test_left_padded_batch_equals_single_runs_f32(tiny):
    """Ragged batching (VERDICT r2 missing #3): prompts of different lengths in ONE generate as a left-padded batch with
    `attention_mask` (what the reference forwards to HF's generate, modeling_gar.py:418-426) give, token for token, what
    each prompt gives alone and unpadded — the single runs being the oracle-checked path; logits within the f32 tolerance
    (the kv tiles of a shifted sequence sum in another order). Image path (placeholders and crop spans move with the
    padding) and text-only path against the oracle's padded batch (pinned on transformers, tests/test_oracle_goldens.py)."""
    from gar_amd.modeling_gar import GARModel
    from oracle import gar_oracle as O
    cfg, W, proc = tiny
    base = [_sample(cfg, proc, i) for i in (3, 4, 6, 7)]
    ss = [_lengthen(x, e, 100 + i) for i, (x, e) in enumerate(zip(base, (0, 5, 37, 130)))]
    assert len({x["input_ids"].shape[1] for x in ss}) == 4
    m = GARModel(cfg, W, torch.float32)
    n = 10
    singles = [m.generate(**x, max_new_tokens=n, return_logits=True) for x in ss]
    for x, o in zip(ss, singles):               # the single runs are the oracle's
        ref_seq, ref_logits = _oracle(W, cfg, x, n)
        _check_f32(o, ref_seq, ref_logits, "single")
    batch = _left_pad_batch(ss)
    for use_graph in (True, False):
        out = m.generate(**batch, max_new_tokens=n, return_logits=True, use_graph=use_graph)
        for b, o in enumerate(singles):
            assert out.sequences[b].tolist() == o.sequences[0].tolist(), (b, use_graph)
            err = float((out.logits[b] - o.logits[0]).abs().max())
            assert err <= F32_LOGIT_TOL * float(o.logits.abs().max()), (b, err)
    # the same request again on the cached state, then an unpadded request of the same shape: left_pad must be reset
    same_len = [_lengthen(x, 130, 200 + i) for i, x in enumerate(base[:2])]
    plain = dict(_left_pad_batch(same_len))
    assert bool(plain["attention_mask"].all())
    ref = [m.generate(**x, max_new_tokens=4).sequences[0].tolist() for x in same_len]
    four = _left_pad_batch([ss[0], ss[3]])
    m.generate(**four, max_new_tokens=4)
    assert m.generate(**plain, max_new_tokens=4).sequences.tolist() == ref
    # a right-padded mask is refused (HF would continue the row after its padding)
    bad = dict(batch)
    bad["attention_mask"] = batch["attention_mask"].flip(1)
    with pytest.raises(Exception, match="LEFT-padded"):
        m.generate(**bad, max_new_tokens=2)
    # text-only prompts of three lengths against the oracle's padded batch
    g = torch.Generator().manual_seed(5)
    lens = [40, 23, 9]
    S = max(lens)
    ids = torch.full((3, S), 7, dtype=torch.int64)
    mask = torch.zeros(3, S, dtype=torch.int64)
    for b, L in enumerate(lens):
        ids[b, S - L:] = torch.randint(10, 290, (L,), generator=g)
        mask[b, S - L:] = 1
    tcfg = cfg.mllm_config.text_config
    emb = torch.nn.functional.embedding(ids, W[O.LM + "embed_tokens.weight"])
    ref_seq, ref_logits = O.greedy_generate(emb, W, tcfg, 8, return_logits=True, attention_mask=mask)
    out = m.generate(input_ids=ids, attention_mask=mask, max_new_tokens=8, return_logits=True)
    _check_f32(out, ref_seq, ref_logits, "text-only padded batch")


def test_left_padded_batch_bf16(tiny):
    """bf16: the padded batch against the single runs — first-token logits within the bf16 tolerance of each other (the
    kv tiles of a shifted sequence round in another order), finite everywhere, graph replay == eager, and no validate
    sync needed (the pad is derived from the mask on the device)."""
    from gar_amd.modeling_gar import GARModel
    cfg, W, proc = tiny
    base = [_sample(cfg, proc, i, dtype=torch.bfloat16) for i in (3, 4, 6)]
    ss = [_lengthen(x, e, 100 + i) for i, (x, e) in enumerate(zip(base, (0, 70, 19)))]
    m = GARModel(cfg, W, torch.bfloat16)
    singles = [m.generate(**x, max_new_tokens=6, return_logits=True) for x in ss]
    batch = _left_pad_batch(ss)
    out = m.generate(**batch, max_new_tokens=6, return_logits=True, validate=False)
    assert torch.isfinite(out.logits).all() and int(out.input_flags.item()) == 0
    for b, o in enumerate(singles):
        assert _rel_l2(out.logits[b, 0], o.logits[0, 0]) < 2e-2, b
    eager = m.generate(**batch, max_new_tokens=6, return_logits=True, use_graph=False)
    assert torch.equal(eager.sequences, out.sequences) and torch.equal(eager.logits, out.logits)
