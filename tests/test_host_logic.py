"""CPU: host-side logic (processor, sample builders, config, weights) against outputs of the reference's own
pure-Python helpers executed in the build container (tests/golden/ref_helpers.json, tools/make_goldens.py)."""
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from gar_amd import GARConfig
from gar_amd.eval_dataset import MultiRegionDataset, SingleRegionCaptionDataset
from gar_amd.processing import GARProcessor, StubTokenizer, select_canvas, split_tiles
from gar_amd.weights import check_weights, load_weights, save_weights, synthetic_weights, weight_shapes


@pytest.fixture(scope="module")
def ref(golden_dir):
    return json.load(open(os.path.join(golden_dir, "ref_helpers.json")))


def test_canvas_table_matches_reference(ref):
    for w, h, mt, cw, ch in ref["canvas_table"]:
        assert select_canvas(w, h, 448, mt) == (cw, ch), (w, h, mt)


def test_split_matches_reference(ref):
    r = ref["split_merge"]
    ncw, nch, th, tw, C = r["ncw"], r["nch"], r["th"], r["tw"], r["C"]
    img = torch.arange(C * nch * th * ncw * tw, dtype=torch.float32).view(1, C, nch * th, ncw * tw)
    assert split_tiles(img, ncw, nch).flatten().tolist() == r["tiles"]


def test_single_region_parse_matches_reference(ref, golden_dir):
    d1 = ref["demo1"]
    img = Image.new("RGBA", tuple(d1["size"]))
    mask = np.array(Image.open(os.path.join(golden_dir, "demo_mask_1.png")).convert("L")).astype(bool)
    proc = GARProcessor(StubTokenizer(), max_num_tiles=16)
    ds = SingleRegionCaptionDataset(img, mask, proc, data_dtype=torch.float32, device="cpu")
    d = ds._parse_annotations()
    assert d["visual_prompt"].mode == d1["vp_mode"]
    vals, cnts = np.unique(np.array(d["visual_prompt"]), return_counts=True)
    assert {str(int(v)): int(c) for v, c in zip(vals, cnts)} == d1["hist"]
    assert sorted(int(v) for v in np.unique(np.array(d["visual_prompt"].convert("RGB")))) == d1["rgb_uniques"]
    assert {k: [float(x) for x in v] for k, v in d["bboxes"].items()} == d1["bboxes"]


def test_multi_region_parse_matches_reference_including_quirks(ref, golden_dir):
    d3 = ref["demo3"]
    img = Image.new("RGB", tuple(d3["size"]))
    masks = [np.array(Image.open(os.path.join(golden_dir, f"demo_mask_3_{i}.png")).convert("L")).astype(bool)
             for i in range(3)]
    proc = GARProcessor(StubTokenizer(), max_num_tiles=16)
    ds = MultiRegionDataset(img, masks, d3["question"], proc, data_dtype=torch.float32, device="cpu",
                            prompt_order=d3["order"])
    d = ds._parse_annotations()
    assert d["prompt"] == d3["prompt"]
    vals, cnts = np.unique(np.array(d["visual_prompt"]), return_counts=True)
    assert {str(int(v)): int(c) for v, c in zip(vals, cnts)} == d3["hist"]
    got = {k: [float(x) for x in v] for k, v in d["bboxes"].items()}
    assert got == d3["bboxes"]
    assert len({tuple(v) for v in got.values()}) == 1          # stale mask_id quirk: all boxes = last mask's


def test_sample_contract_full_size():
    """Key names / shapes / dtypes of the generate kwargs (evaluation/eval_dataset.py:141-148) at 1024^2, mt16."""
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    cfg = GARConfig.gar_1b()
    proc = GARProcessor.from_config(cfg, max_num_tiles=16)
    s = SingleRegionCaptionDataset(synthetic_image(0), synthetic_mask(0), proc, data_dtype=torch.bfloat16,
                                   device="cpu")[0]
    assert s["pixel_values"].shape == (17, 3, 448, 448) and s["pixel_values"].dtype == torch.bfloat16
    assert s["global_mask_values"].shape == (17, 3, 448, 448)
    assert s["aspect_ratios"].tolist() == [[4, 4]]
    ids = s["input_ids"]
    assert ids.dtype == torch.int64 and ids.shape[0] == 1
    assert int((ids == 128002).sum()) == 17 * 256 and int((ids == 128005).sum()) == 256
    assert s["attention_mask"].dtype == torch.bfloat16 and s["attention_mask"].shape == ids.shape
    assert list(s["bboxes"][0].keys()) == ["128005"]
    # mask values decode back to {prompt id 1, NO_Prompt 5} exactly, in bf16
    mv = torch.round((s["global_mask_values"] + 1.0) / 2.0 * 255.0).long()
    assert sorted(mv.unique().tolist()) == [1, 5]


def test_video_sample_contract_and_oracle_replay():
    """A13 builder + oracle: F frames -> F single tiles, F*P*P image placeholders, one crop run and bbox per frame;
    the oracle's video replay writes convex blends of the frame's own pooled tokens."""
    from gar_amd.eval_dataset import VideoRegionCaptionDataset
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    from oracle import gar_oracle as O
    cfg = GARConfig.tiny()
    proc = GARProcessor.from_config(cfg, max_num_tiles=4)
    F_ = 6
    frames = [synthetic_image(f, 150, 120) for f in range(F_)]
    masks = [synthetic_mask(f, 150, 120) for f in range(F_)]
    s = VideoRegionCaptionDataset(frames, masks, proc, data_dtype=torch.float32, device="cpu")[0]
    P2 = cfg.tokens_per_tile
    assert s["pixel_values"].shape == (F_, 3, 112, 112) and s["global_mask_values"].shape == (F_, 3, 112, 112)
    assert int((s["input_ids"] == cfg.mllm_config.image_token_id).sum()) == F_ * P2
    assert s["video_frame_tokens"] == [304, 305, 308, 310, 311, 312]
    for t in s["video_frame_tokens"]:
        assert int((s["input_ids"] == t).sum()) == P2 and str(t) in s["bboxes"][0]
    C = 4
    feats = torch.zeros(F_, P2, C)
    for f in range(F_):
        feats[f, :, 0] = float(f + 1)                 # constant map per frame -> replay rows equal f+1
    emb = torch.zeros(1, s["input_ids"].shape[1], C)
    out = O.feature_replay_video(emb, s["input_ids"], feats, s["bboxes"], s["video_frame_tokens"], cfg)
    for f, t in enumerate(s["video_frame_tokens"]):
        rows = out[0][s["input_ids"][0] == t]
        assert torch.allclose(rows[:, 0], torch.full((P2,), float(f + 1)), atol=1e-6)


def test_tokenizer_roundtrip_and_special_ids():
    tk = StubTokenizer()
    assert tk.convert_tokens_to_ids("<|reserved_special_token_3|>") == 128005
    assert [tk.convert_tokens_to_ids(f"<|reserved_special_token_{k + 2}|>") for k in range(5)] == \
        GARConfig.gar_1b().crop_tokens_ids
    assert tk.convert_tokens_to_ids("<NO_Prompt>") - 128256 == 5
    text = "<|begin_of_text|>héllo <Prompt1>: <|image|>x"
    ids = tk.encode(text)
    assert tk.decode(ids) == text and tk.decode(ids, skip_special_tokens=True) == "héllo : x"


def test_config_contract():
    c = GARConfig.gar_1b()
    assert c.prompt_numbers == 5 and c.kernel_size == [14, 14] and c.mask_path_embedding_out_channels == 1024
    assert c.pooled_side == 16 and c.tokens_per_tile == 256 and c.feat_stride == 28
    c8 = GARConfig.gar_8b()
    assert c8.mask_path_embedding_out_channels == 1536 and not c8.mllm_config.vision_use_cls_token
    assert c8.mllm_config.text_config.head_dim == 128 and c8.mllm_config.vision_config.mlp_dim == 8960
    rt = GARConfig.from_dict(json.loads(json.dumps(c.to_dict())))
    assert rt.to_dict() == c.to_dict()
    with pytest.raises(AssertionError):
        GARConfig(prompt_numbers=4)


def test_weights_naming_and_io(tmp_path):
    cfg = GARConfig.tiny()
    W = synthetic_weights(cfg)
    assert "mllm.lm_head.weight" not in W                       # tied
    assert W["mllm.model.vision_tower.timm_model.pos_embed"].shape == (1, 65, 128)
    p = str(tmp_path / "w.safetensors")
    save_weights(W, p)
    W2 = load_weights(p)
    check_weights(cfg, W2)
    assert all(torch.equal(W[k], W2[k]) for k in W)
    W3 = synthetic_weights(cfg)
    assert all(torch.equal(W[k], W3[k]) for k in W)             # deterministic per name
    del W2["mask_patch_embedding.weight"]
    with pytest.raises(KeyError):
        check_weights(cfg, W2)
    n1b = sum(int(np.prod(s)) for s in weight_shapes(GARConfig.gar_1b()).values())
    assert 1.5e9 < n1b < 1.6e9


@pytest.mark.parametrize("vision_bias", ["split", "unfused"])
def test_checkpoint_normalisation_fills_config_and_folds_bias_variants(vision_bias, hf_style_checkpoint):
    from gar_amd.weights import normalize_checkpoint
    cfg = GARConfig.tiny()
    W = synthetic_weights(cfg)
    ck, d = hf_style_checkpoint(cfg, W, vision_bias)
    cfg2 = GARConfig.from_dict(json.loads(json.dumps(d)))
    assert "depth" not in cfg2.mllm_config.vision_config.model_args
    W2 = normalize_checkpoint(cfg2, ck)
    v = cfg2.mllm_config.vision_config
    assert (v.depth, v.mlp_dim) == (2, 256)
    check_weights(cfg2, W2)
    assert all(torch.equal(W[k], W2[k]) for k in W)
    # a config that disagrees with the tensors is an error, not a silent reshape
    cfg3 = GARConfig.tiny(**{"vision.depth": 3})
    with pytest.raises(ValueError):
        normalize_checkpoint(cfg3, ck)
    # a missing k_bias buffer means zero (timm registers it non-persistent)
    if vision_bias == "split":
        ck = {k: t for k, t in ck.items() if not k.endswith("k_bias")}
        W4 = normalize_checkpoint(GARConfig.from_dict(json.loads(json.dumps(d))), ck)
        b = W4["mllm.model.vision_tower.timm_model.blocks.0.attn.qkv.bias"]
        assert torch.count_nonzero(b[128:256]) == 0 and torch.equal(b[:128], W["mllm.model.vision_tower.timm_model.blocks.0.attn.qkv.bias"][:128])


def test_compact_sincos_table_for_the_fused_rope_epilogue():
    """ops._compact_sincos: (sin, cos)-pair table only when every angle is repeated for both elements of a rotated pair
    (timm's cat layout), with the identity row (0, 1) in front that un-rotated rows / columns read (gar_hip.h, ABI 14); anything
    else keeps the full tables (the library then takes its general epilogue)."""
    from gar_amd import ops
    ang = torch.randn(40, 32)
    sin, cos = ang.sin().repeat_interleave(2, -1).contiguous(), ang.cos().repeat_interleave(2, -1).contiguous()
    sc = ops._compact_sincos(sin, cos)
    assert sc.shape == (41, 32, 2) and sc.is_contiguous()
    assert torch.equal(sc[0, :, 0], torch.zeros(32)) and torch.equal(sc[0, :, 1], torch.ones(32))
    assert torch.equal(sc[1:, :, 0], ang.sin()) and torch.equal(sc[1:, :, 1], ang.cos())
    assert ops._compact_sincos(sin, cos) is sc                              # cached per table pair
    sin2 = sin.clone()
    sin2[3, 5] += 0.25                                                      # pair no longer shares its angle
    assert ops._compact_sincos(sin2, cos) is None


def test_processor_with_a_real_hf_tokenizer_directory(tmp_path):
    """GARProcessor.from_pretrained on a directory holding a real `tokenizers` fast tokenizer (byte-level BPE without
    merges, the Llama-3 / PLM special tokens at their released ids — built here, there is no hub access): the HF adapter
    path gives the same sample as the stub tokenizer: same length, same special-token positions and ids (image-token
    expansion, crop tokens, chat template), same text, identical pixel / mask tensors and boxes."""
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from transformers import PreTrainedTokenizerFast
    from gar_amd.processing import LLAMA3_SPECIALS
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    vocab = {ch: i for i, ch in enumerate(sorted(pre_tokenizers.ByteLevel.alphabet()))}
    inv = {v: k for k, v in LLAMA3_SPECIALS.items()}
    for i in range(256, max(LLAMA3_SPECIALS.values()) + 1):
        vocab[inv.get(i, f"<|filler_{i}|>")] = i
    tk = Tokenizer(models.BPE(vocab=vocab, merges=[]))
    tk.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tk.decoder = decoders.ByteLevel()
    tk.add_special_tokens(list(LLAMA3_SPECIALS.keys()))
    PreTrainedTokenizerFast(tokenizer_object=tk, eos_token="<|eot_id|>", pad_token="<|end_of_text|>",
                            bos_token="<|begin_of_text|>").save_pretrained(str(tmp_path))
    cfg = GARConfig.gar_1b()
    ph = GARProcessor.from_pretrained(str(tmp_path), cfg, max_num_tiles=4)
    ps = GARProcessor.from_config(cfg, max_num_tiles=4)
    assert type(ph.tokenizer).__name__ == "_HFTokenizerAdapter"
    assert (ph.tokenizer.image_token_id, ph.tokenizer.eos_token_id) == (128002, 128009)
    img, m = synthetic_image(1, 200, 160), synthetic_mask(1, 200, 160)
    a = SingleRegionCaptionDataset(img, m, ph, data_dtype=torch.float32, device="cpu")[0]
    b = SingleRegionCaptionDataset(img, m, ps, data_dtype=torch.float32, device="cpu")[0]
    ia, ib = a["input_ids"][0], b["input_ids"][0]
    assert ia.shape == ib.shape and int((ia == 128002).sum()) == int((ib == 128002).sum()) > 0
    sa, sb = ia >= 128000, ib >= 128000
    assert torch.equal(sa, sb) and torch.equal(ia[sa], ib[sb])
    assert ph.tokenizer.decode(ia[~sa].tolist()) == ps.tokenizer.decode(ib[~sb].tolist())
    assert torch.equal(a["pixel_values"], b["pixel_values"]) and torch.equal(a["global_mask_values"], b["global_mask_values"])
    assert a["bboxes"] == b["bboxes"]


def test_resampling_tables_for_gpu_preprocessing():
    """tap tables used by the device preprocessing: weights reproduce torch's antialiased bicubic exactly (a resize
    done with the tables in numpy, sequential fma order, equals F.interpolate), NEAREST indices = floor(i * in/out)."""
    import numpy as np
    import torch.nn.functional as F
    from gar_amd.preprocess_gpu import bicubic_aa_taps, nearest_index
    rng = np.random.default_rng(0)
    for n_in, n_out in [(53, 24), (48, 96), (100, 64)]:
        first, count, w = bicubic_aa_taps(n_in, n_out)
        assert np.allclose(w.sum(1), 1.0, atol=1e-5) and (count > 0).all() and (first + count <= n_in).all()
        x = rng.integers(0, 256, (5, n_in)).astype(np.float32)
        ref = F.interpolate(torch.from_numpy(x).reshape(1, 1, 5, n_in), size=(5, n_out), mode="bicubic",
                            align_corners=False, antialias=True)[0, 0].numpy()
        out = np.zeros((5, n_out), np.float32)
        for i in range(n_out):
            t = (x[:, first[i]] * w[i, 0]).astype(np.float32)
            for j in range(1, count[i]):
                t = (x[:, first[i] + j].astype(np.float64) * np.float64(w[i, j]) + t.astype(np.float64)).astype(np.float32)
            out[:, i] = t
        assert np.abs(out - ref).max() <= 1e-4          # exact on FMA hosts; unfused hosts differ in the last bits
    for n_in, n_out in [(1024, 1792), (770, 448), (5, 448)]:
        idx = nearest_index(n_in, n_out)
        scale = np.float32(n_in) / np.float32(n_out)
        exp = np.minimum(np.floor(np.arange(n_out, dtype=np.float32) * scale).astype(np.int32), n_in - 1)
        assert np.array_equal(idx, exp)


def test_coco_rle_against_reference_annotation_samples(golden_dir):
    """gar_rle_decode (host C++ in libgar_hip.so) on run-length masks copied from the reference's benchmark annotation
    files: the runs cover exactly h*w pixels, the decoded object sits where that file's own bbox says (objects365 boxes are
    loose: box IoU >= 0.6; a transposed or row-major decode gives ~0), re-encoding reproduces the file's `counts` string byte for byte, and random masks round-trip
    (incl. empty / full / single-pixel, non-square)."""
    import json
    import numpy as np
    from gar_amd import rle
    d = json.load(open(os.path.join(golden_dir, "rle_samples.json")))
    for a in d["dlc"]:
        m = rle.decode(a["segmentation"])
        assert list(m.shape) == a["image_hw"] and set(np.unique(m)) <= {0, 1}
        ys, xs = np.nonzero(m)
        gx, gy, gw, gh = xs.min(), ys.min(), xs.max() - xs.min() + 1, ys.max() - ys.min() + 1
        bx, by, bw, bh = a["bbox"]
        iw = min(gx + gw, bx + bw) - max(gx, bx)
        ih = min(gy + gh, by + bh) - max(gy, by)
        inter = max(iw, 0) * max(ih, 0)
        assert inter / (gw * gh + bw * bh - inter) >= 0.6, ((gx, gy, gw, gh), a["bbox"])
        assert rle.encode(m)["counts"] == a["segmentation"]["counts"]
    for it in d["gar_bench"]:
        for r in it["mask_rles"]:
            m = rle.decode(r)
            assert m.sum() > 0 and rle.encode(m)["counts"] == r["counts"]
    rng = np.random.default_rng(0)
    cases = [rng.random((h, w)) < p for h, w, p in [(7, 5, 0.5), (64, 48, 0.3), (33, 129, 0.02), (1, 9, 0.5), (9, 1, 0.5)]]
    cases += [np.zeros((5, 6), bool), np.ones((5, 6), bool), np.eye(4, dtype=bool)]
    for m in cases:
        e = rle.encode(m)
        assert np.array_equal(rle.decode(e).astype(bool), m)
        runs = []                                    # uncompressed form of the same mask decodes identically
        flat = m.T.reshape(-1)
        cur, n = False, 0
        for v in flat:
            if bool(v) == cur:
                n += 1
            else:
                runs.append(n)
                cur, n = bool(v), 1
        runs.append(n)
        assert np.array_equal(rle.decode({"size": list(m.shape), "counts": runs}).astype(bool), m)
    import pytest
    from gar_amd import hip
    with pytest.raises(hip.GarError):
        rle.decode({"size": [4, 4], "counts": "3"})          # covers 3 of 16 pixels


def test_polygon_rasteriser_known_answers():
    """rle.from_polygons (COCO frPyObjects + merge + decode, restated — parity unpinned, pycocotools is not available):
    axis-aligned rectangles of the kind the reference's Ferret-Bench annotations hold cover exactly their pixel box,
    a right triangle has its analytic area, the union of overlapping polygons is a union, and clipping at the image
    border works."""
    import numpy as np
    from gar_amd import rle
    m = rle.from_polygons([[230.39, 52.48, 286.41, 52.48, 286.41, 84.48, 230.39, 84.48]], 480, 640)   # annotation 0
    ys, xs = np.nonzero(m)
    assert (xs.min(), xs.max(), ys.min(), ys.max()) == (230, 285, 52, 83) and m.sum() == 56 * 32
    t = rle.from_polygons([[10.0, 10.0, 110.0, 10.0, 10.0, 60.0]], 100, 150)
    assert t.sum() == 2500 and t[10, 10] == 1 and t[59, 10] == 1 and t[59, 30] == 0
    a = [10.0, 10.0, 50.0, 10.0, 50.0, 50.0, 10.0, 50.0]
    b = [30.0, 30.0, 70.0, 30.0, 70.0, 70.0, 30.0, 70.0]
    u = rle.from_polygons([a, b], 100, 100)
    assert u.sum() == 1600 + 1600 - 400
    c = rle.from_polygons([[-20.0, -20.0, 30.0, -20.0, 30.0, 30.0, -20.0, 30.0]], 64, 64)
    assert c.sum() == 30 * 30 and c[0, 0] == 1


def test_data_type_choices_are_the_reference_clis():
    """The reference CLIs offer --data_type fp16 | bf16 | fp32 (demo/gar_with_mask.py:44): each maps to its own arithmetic —
    fp16 to the twin library (hip.lib(torch.float16)), never silently to bf16 — and the twin binds the same ABI."""
    from gar_amd import hip
    from gar_amd.bench_loops import DATA_TYPE_CHOICES, resolve_data_type
    assert DATA_TYPE_CHOICES == ["fp16", "bf16", "fp32"]
    assert resolve_data_type("bf16") is torch.bfloat16 and resolve_data_type("fp32") is torch.float32
    assert resolve_data_type("fp16") is torch.float16
    assert hip.dtype_code(torch.float16) == hip.GAR_BF16 == hip.dtype_code(torch.bfloat16)      # "the library's 16-bit type"
    a, b = hip.lib(torch.bfloat16), hip.lib(torch.float16)
    assert a is not b and a is hip.lib(torch.float32) and a.gar_abi_version() == b.gar_abi_version() == hip.ABI_VERSION


def test_chat_template_of_a_checkpoint_directory_is_honoured(tmp_path):
    """(f1) readiness: a hub snapshot carries its chat template (chat_template.jinja / chat_template.json / processor_config.json /
    tokenizer_config.json, transformers' precedence); GARProcessor.from_pretrained renders THAT through transformers' own Jinja
    environment. The PLM template gives exactly the built-in layout; another template is honoured, not overridden."""
    import json
    from gar_amd import GARConfig
    from gar_amd.processing import GARProcessor
    plm = ("{{- bos_token }}{%- for message in messages %}{{- '<|start_header_id|>' + message['role'] + '<|end_header_id|>\\n\\n' }}"
           "{%- if message['content'] is string %}{{- message['content'] }}{%- else %}{%- for content in message['content'] %}"
           "{%- if content['type'] == 'image' %}{{- '<|image|>' }}{%- elif content['type'] == 'text' %}{{- content['text'] }}{%- endif %}"
           "{%- endfor %}{%- endif %}{{- '<|eot_id|>' }}{%- endfor %}"
           "{%- if add_generation_prompt %}{{- '<|start_header_id|>assistant<|end_header_id|>\\n\\n' }}{%- endif %}")
    msgs = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "Describe <Prompt0> in detail."}]},
            {"role": "assistant", "content": "A red kite."},
            {"role": "user", "content": [{"type": "text", "text": "And <Prompt1>?"}]}]
    cfg = GARConfig.tiny()
    builtin = GARProcessor.from_config(cfg)
    d1 = tmp_path / "a"
    d1.mkdir()
    (d1 / "chat_template.jinja").write_text(plm)
    p1 = GARProcessor.from_pretrained(str(d1), cfg)
    assert p1.chat_template == plm
    for gen in (True, False):
        assert p1.apply_chat_template(msgs, add_generation_prompt=gen) == builtin.apply_chat_template(msgs, add_generation_prompt=gen)
    d2 = tmp_path / "b"
    d2.mkdir()
    other = "{{- bos_token }}SYSTEM: be brief." + plm.replace("{{- bos_token }}", "", 1)
    (d2 / "chat_template.json").write_text(json.dumps({"chat_template": other}))
    (d2 / "tokenizer_config.json").write_text(json.dumps({"chat_template": "ignored: lower precedence"}))
    p2 = GARProcessor.from_pretrained(str(d2), cfg)
    out = p2.apply_chat_template(msgs)
    assert out.startswith("<|begin_of_text|>SYSTEM: be brief.<|start_header_id|>user") and out.endswith("assistant<|end_header_id|>\n\n")
    d3 = tmp_path / "c"
    d3.mkdir()
    assert GARProcessor.from_pretrained(str(d3), cfg).chat_template is None


def test_chat_template_renders_without_transformers_private_compiler(tmp_path, monkeypatch):
    """apply_chat_template used a private transformers symbol; on a version without it the template is compiled by the
    Jinja sandbox built in gar_amd.processing — same rendering, and the compiled template is cached."""
    import sys
    import types
    from gar_amd import GARConfig
    from gar_amd.processing import GARProcessor
    cfg = GARConfig.tiny()
    plm = ("{{- bos_token }}{%- for message in messages %}{{- '<|start_header_id|>' + message['role'] + '<|end_header_id|>\n\n' }}"
           "{%- for content in message['content'] %}{%- if content['type'] == 'image' %}{{- '<|image|>' }}"
           "{%- elif content['type'] == 'text' %}{{- content['text'] }}{%- endif %}{%- endfor %}{{- '<|eot_id|>' }}{%- endfor %}"
           "{%- if add_generation_prompt %}{{- '<|start_header_id|>assistant<|end_header_id|>\n\n' }}{%- endif %}")
    d = tmp_path / "ckpt"
    d.mkdir()
    (d / "chat_template.jinja").write_text(plm)
    msgs = [{"role": "user", "content": [{"type": "image"}, {"type": "text", "text": "Describe the masked region."}]}]
    p = GARProcessor.from_pretrained(str(d), cfg)
    with_private = p.apply_chat_template(msgs)
    stub = types.ModuleType("transformers.utils.chat_template_utils")       # a transformers without _compile_jinja_template
    monkeypatch.setitem(sys.modules, "transformers.utils.chat_template_utils", stub)
    p2 = GARProcessor.from_pretrained(str(d), cfg)
    assert p2.apply_chat_template(msgs) == with_private == GARProcessor.from_config(cfg).apply_chat_template(msgs)
    first = p2._chat_template_compiled[1]
    p2.apply_chat_template(msgs)
    assert p2._chat_template_compiled[1] is first
