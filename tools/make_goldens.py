#!/usr/bin/env python
"""Generates tests/golden/*  — runs ONLY in the build container (needs /root/reference and transformers).

Nothing from the reference is copied: this script imports / executes pieces of the reference (and of the
third-party packages the reference calls) to produce *data* — inputs and expected outputs — that pin the
oracle (oracle/gar_oracle.py) and the host logic (gar_amd.processing / gar_amd.eval_dataset).

Fixtures written:
  llama_tiny.npz        transformers LlamaForCausalLM (eager, fp32): logits of a prefill + greedy tokens from
                        inputs_embeds, llama3 rope scaling, GQA, tied head          -> pins oracle.llama_* / greedy
  llama_tiny_padded.npz a left-padded batch (23 / 17 / 9 embeddings) through the same model's generate(inputs_embeds=,
                        attention_mask=): tokens + per-step scores                   -> pins the oracle's attention_mask path
  projector_tiny.npz    transformers PerceptionLMMultiModalProjector (+AdaptiveAvgPooling) -> pins projector_forward
  torch_ops.npz         torch conv2d / layer_norm / SDPA / GELU building blocks the ViT restatement uses
  ref_helpers.json      outputs of the reference's own pure-Python helpers executed here:
                          canvas selection table (image_processing_perception_lm_fast.py:95-252),
                          _merge / _split index maps (modeling_gar.py:248-260, image_processing...:254-266),
                          eval_dataset._parse_annotations on the demo assets (evaluation/eval_dataset.py:56-99,192-260)
  demo_mask_*.png       the reference's demo masks (data files, assets/), inputs of the eval_dataset goldens
"""
import ast
import json
import os
import shutil
import sys
import textwrap

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))


def _extract(path, cls, names):
    """Source of selected methods of ``cls`` in ``path`` as plain functions (dedented)."""
    src = open(path).read()
    tree = ast.parse(src)
    out = {}
    for node in tree.body:
        if isinstance(node, ast.ClassDef) and node.name == cls:
            for f in node.body:
                if isinstance(f, ast.FunctionDef) and f.name in names:
                    seg = ast.get_source_segment(src, f)
                    seg = "\n".join(l for l in textwrap.dedent(seg).split("\n") if not l.strip().startswith("@"))
                    out[f.name] = seg
    return out


def golden_llama():
    from transformers import LlamaConfig, LlamaForCausalLM
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, vocab_size=512, rms_norm_eps=1e-5, rope_theta=500000.0,
                      rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                    "original_max_position_embeddings": 8192, "rope_type": "llama3"},
                      max_position_embeddings=131072, tie_word_embeddings=True, attention_bias=False,
                      mlp_bias=False, attn_implementation="eager")
    m = LlamaForCausalLM(cfg).eval().to(torch.float32)
    with torch.no_grad():
        for n, p in m.named_parameters():
            if p.dim() == 2 and "embed" not in n:
                p.copy_(torch.randn_like(p) / (p.shape[1] ** 0.5))
            elif "embed" in n:
                p.copy_(torch.randn_like(p) * 0.05)
            else:
                p.copy_(1.0 + 0.05 * torch.randn_like(p))
    S = 23
    # long positions exercise the llama3 low-frequency rescaling
    emb = torch.randn(1, S, 128)
    with torch.no_grad():
        logits = m(inputs_embeds=emb).logits
        gen = m.generate(inputs_embeds=emb, attention_mask=torch.ones(1, S, dtype=torch.long), max_new_tokens=12,
                         do_sample=False, use_cache=True, return_dict_in_generate=True, pad_token_id=0,
                         eos_token_id=None)
    W = {}
    for n, p in m.state_dict().items():
        if n == "lm_head.weight":
            continue
        W["mllm.model.language_model." + n[len("model."):]] = p.numpy()
    from transformers.modeling_rope_utils import ROPE_INIT_FUNCTIONS
    inv, _ = ROPE_INIT_FUNCTIONS["llama3"](cfg, "cpu")
    np.savez_compressed(os.path.join(OUT, "llama_tiny.npz"), inputs_embeds=emb.numpy(), logits=logits.numpy(),
                        sequences=gen.sequences.numpy(), inv_freq=inv.numpy(),
                        **{"W:" + k: v for k, v in W.items()})
    # a second inv_freq vector at the real GAR-1B / 8B head dims
    extra = {}
    for name, hd, factor in (("1b", 64, 32.0), ("8b", 128, 8.0)):
        c = LlamaConfig(hidden_size=hd * 32, num_attention_heads=32, head_dim=hd, rope_theta=500000.0,
                        rope_scaling={"factor": factor, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                      "original_max_position_embeddings": 8192, "rope_type": "llama3"})
        extra["inv_freq_" + name] = ROPE_INIT_FUNCTIONS["llama3"](c, "cpu")[0].numpy()
    np.savez_compressed(os.path.join(OUT, "llama_inv_freq.npz"), **extra)
    print("llama_tiny: tokens", gen.sequences.tolist())


def golden_llama_padded():
    """A LEFT-PADDED batch through transformers' LlamaForCausalLM.generate(inputs_embeds=, attention_mask=) — the call
    GARModel.generate ends in (modeling_gar.py:418-426) — on the model of llama_tiny.npz: three prompts of 23 / 17 / 9
    embeddings padded to 23, 10 greedy tokens, per-step scores. Pins the oracle's attention_mask handling (position ids
    from the mask, padding keys hidden)."""
    from transformers import LlamaConfig, LlamaForCausalLM
    g = np.load(os.path.join(OUT, "llama_tiny.npz"))
    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_hidden_layers=2, num_attention_heads=4,
                      num_key_value_heads=2, head_dim=64, vocab_size=512, rms_norm_eps=1e-5, rope_theta=500000.0,
                      rope_scaling={"factor": 32.0, "low_freq_factor": 1.0, "high_freq_factor": 4.0,
                                    "original_max_position_embeddings": 8192, "rope_type": "llama3"},
                      max_position_embeddings=131072, tie_word_embeddings=True, attention_bias=False,
                      mlp_bias=False, attn_implementation="eager")
    m = LlamaForCausalLM(cfg).eval().to(torch.float32)
    sd = {"model." + k[2:][len("mllm.model.language_model."):]: torch.from_numpy(g[k]) for k in g.files if k.startswith("W:")}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    m.load_state_dict(sd)
    torch.manual_seed(7)
    S, lens = 23, [23, 17, 9]
    emb = torch.randn(3, S, 128)
    mask = torch.zeros(3, S, dtype=torch.long)
    for b, n in enumerate(lens):
        mask[b, S - n:] = 1
    with torch.no_grad():
        gen = m.generate(inputs_embeds=emb, attention_mask=mask, max_new_tokens=10, do_sample=False, use_cache=True,
                         return_dict_in_generate=True, output_scores=True, pad_token_id=0, eos_token_id=None)
        # the same prompts one at a time, unpadded: what a padded row must reproduce
        singles = [m.generate(inputs_embeds=emb[b:b + 1, S - n:], attention_mask=torch.ones(1, n, dtype=torch.long),
                              max_new_tokens=10, do_sample=False, use_cache=True, return_dict_in_generate=True,
                              pad_token_id=0, eos_token_id=None).sequences[0] for b, n in enumerate(lens)]
    np.savez_compressed(os.path.join(OUT, "llama_tiny_padded.npz"), inputs_embeds=emb.numpy(), attention_mask=mask.numpy(),
                        sequences=gen.sequences.numpy(), scores=torch.stack(gen.scores, 1).numpy(),
                        single_sequences=torch.stack(singles).numpy())
    print("llama_tiny_padded: tokens", gen.sequences.tolist(), "singles", torch.stack(singles).tolist())


def golden_projector():
    from transformers import PerceptionLMConfig
    from transformers.models.perception_lm.modeling_perception_lm import PerceptionLMMultiModalProjector
    torch.manual_seed(1)
    cfg = PerceptionLMConfig(
        vision_config={"model_type": "timm_wrapper", "architecture": "vit_pe_lang_tiny",
                       "model_args": {"embed_dim": 96, "img_size": [112, 112], "ref_feat_shape": [8, 8]}},
        text_config={"model_type": "llama", "hidden_size": 128, "intermediate_size": 256, "num_hidden_layers": 1,
                     "num_attention_heads": 2, "vocab_size": 64},
        projector_pooling_ratio=2)
    pj = PerceptionLMMultiModalProjector(cfg).eval().to(torch.float32)
    x = torch.randn(3, 64, 96)
    with torch.no_grad():
        y = pj(x)
    np.savez_compressed(os.path.join(OUT, "projector_tiny.npz"), x=x.numpy(), y=y.numpy(),
                        w1=pj.linear_1.weight.detach().numpy(), b1=pj.linear_1.bias.detach().numpy(),
                        w2=pj.linear_2.weight.detach().numpy(), b2=pj.linear_2.bias.detach().numpy())
    print("projector_tiny:", tuple(y.shape))


def golden_ref_helpers():
    import re
    from functools import reduce
    import math
    from PIL import Image
    res = {}

    # ---- canvas selection: execute the reference's methods on a bare object ----------------------------------
    names = ["_factors", "_find_supported_aspect_ratios", "_get_image_height_width", "_fit_image_to_canvas",
             "_find_closest_aspect_ratio", "_split"]
    src = _extract(f"{REF}/projects/grasp_any_region/models/modeling/image_processing_perception_lm_fast.py",
                   "PerceptionLMImageProcessorFast", names)
    ns = {"reduce": reduce, "math": math, "torch": torch}
    for k in names:
        exec(src[k], ns)

    class IP:
        pass
    for k in names:
        setattr(IP, k, staticmethod(ns[k]) if k == "_factors" else ns[k])
    table = []
    sizes = [(1024, 1024), (1024, 770), (640, 427), (2048, 1365), (448, 448), (300, 1200), (1200, 300),
             (1792, 1792), (1793, 900), (5000, 3000), (333, 777), (449, 449), (100, 100)]
    for mt in (1, 4, 8, 16, 36):
        ip = IP()
        ip.max_num_tiles = mt
        for (w, h) in sizes:
            if mt > 1:
                c = ip._fit_image_to_canvas(img_width=w, img_height=h, tile_size=448)
                if c is None:
                    c = ip._find_closest_aspect_ratio(img_width=w, img_height=h, tile_size=448)
            else:
                c = (1, 1)
            table.append([w, h, mt, int(c[0]), int(c[1])])
    res["canvas_table"] = table

    # ---- _split / _merge index maps ------------------------------------------------------------------------
    msrc = _extract(f"{REF}/projects/grasp_any_region/hf_models/modeling_gar.py", "GARModel", ["_merge"])
    exec(msrc["_merge"], ns)
    ncw, nch, th, tw, C = 3, 2, 4, 4, 2
    img = torch.arange(1 * C * nch * th * ncw * tw, dtype=torch.float32).view(1, C, nch * th, ncw * tw)
    tiles = ns["_split"](None, img, ncw, nch)
    merged = ns["_merge"](None, tiles, ncw, nch)
    res["split_merge"] = {"ncw": ncw, "nch": nch, "th": th, "tw": tw, "C": C,
                          "tiles": tiles.flatten().tolist(), "roundtrip_equal": bool(torch.equal(merged, img))}

    # ---- eval_dataset._parse_annotations on the demo assets ---------------------------------------------------
    from gar_amd.processing import StubTokenizer
    esrc = open(f"{REF}/evaluation/eval_dataset.py").read()

    class _NP:  # numpy 1.26 semantics of  -1 * np.ones(shape, uint8)  -> int16   
        def __getattr__(self, k):
            return getattr(np, k)

        @staticmethod
        def ones(shape, dtype=None):
            return np.ones(shape, dtype=np.int16 if dtype is np.uint8 else dtype)
    ens = {}
    mod = compile(esrc, "eval_dataset_ref", "exec")
    exec(mod, ens)
    ens["np"] = _NP()

    class P:
        tokenizer = StubTokenizer()
    img1 = Image.open(f"{REF}/assets/demo_image_1.png")
    m1 = np.array(Image.open(f"{REF}/assets/demo_mask_1.png").convert("L")).astype(bool)
    ds = ens["SingleRegionCaptionDataset"](image=img1, mask=m1, processor=P())
    d = ds._parse_annotations()
    vp = np.array(d["visual_prompt"])
    vals, cnts = np.unique(vp, return_counts=True)
    rgb = np.array(d["visual_prompt"].convert("RGB"))
    res["demo1"] = {"size": list(img1.size), "vp_mode": d["visual_prompt"].mode,
                    "hist": {int(v): int(c) for v, c in zip(vals, cnts)},
                    "rgb_uniques": sorted(int(v) for v in np.unique(rgb)),
                    "bboxes": {k: [float(x) for x in v] for k, v in d["bboxes"].items()}}
    img3 = Image.open(f"{REF}/assets/demo_image_3.png")
    masks3 = [np.array(Image.open(f"{REF}/assets/demo_mask_3_{i}.png").convert("L")).astype(bool) for i in range(3)]
    q = ("Question: What is the relationship between <Prompt0>, <Prompt1>, and <Prompt2>?\nOptions:\n"
         "A. <Prompt0> is wearing <Prompt1>\nB. <Prompt0> is holding <Prompt2>")
    md = ens["MultiRegionDataset"](image=img3, masks=masks3, question_str=q, processor=P())
    dd = md._parse_annotations()
    order = re.findall(r"(<Prompt\d+>); ", dd["prompt"].split("\n")[0])
    vp3 = np.array(dd["visual_prompt"])
    vals, cnts = np.unique(vp3, return_counts=True)
    res["demo3"] = {"size": list(img3.size), "question": q, "order": order, "prompt": dd["prompt"],
                    "hist": {int(v): int(c) for v, c in zip(vals, cnts)},
                    "bboxes": {k: [float(x) for x in v] for k, v in dd["bboxes"].items()}}
    for f in ("demo_mask_1.png", "demo_mask_3_0.png", "demo_mask_3_1.png", "demo_mask_3_2.png"):
        shutil.copyfile(f"{REF}/assets/{f}", os.path.join(OUT, f))
    with open(os.path.join(OUT, "ref_helpers.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("ref_helpers: demo1 bbox", res["demo1"]["bboxes"], "demo3 order", order)


def golden_torch_ops():
    """Building blocks of the ViT restatement evaluated by torch's own modules (not by oracle code)."""
    import torch.nn as nn
    torch.manual_seed(2)
    conv = nn.Conv2d(3, 32, 14, 14, bias=False)
    x = torch.randn(2, 3, 28, 42)
    ln = nn.LayerNorm(32, eps=1e-5)
    with torch.no_grad():
        ln.weight.copy_(1 + 0.1 * torch.randn(32))
        ln.bias.copy_(0.1 * torch.randn(32))
        y = conv(x)
        t = y.flatten(2).transpose(1, 2)
        z = ln(t)
        g = nn.GELU()(z)
        q, k, v = torch.randn(3, 2, 4, 9, 16).unbind(0)
        a = torch.nn.functional.scaled_dot_product_attention(q, k, v)
    np.savez_compressed(os.path.join(OUT, "torch_ops.npz"), x=x.numpy(), conv_w=conv.weight.detach().numpy(),
                        conv_y=y.numpy(), ln_w=ln.weight.detach().numpy(), ln_b=ln.bias.detach().numpy(),
                        ln_y=z.numpy(), gelu_y=g.numpy(), q=q.numpy(), k=k.numpy(), v=v.numpy(), sdpa=a.numpy())


def golden_rle_samples():
    """COCO run-length masks the reference's benchmark annotation files hold (data): a few `segmentation` entries of
    evaluation/DLC-Bench/annotations/annotations.json together with that file's own `bbox` / image size for the same
    object, and a few `mask_rles` of evaluation/GAR-Bench/annotations/GAR-Bench-VQA.json."""
    import ast
    import json
    ref = os.environ.get("GAR_REFERENCE", "/root/reference")
    d = json.load(open(os.path.join(ref, "evaluation/DLC-Bench/annotations/annotations.json")))
    imgs = {str(i["id"]): i for i in d["images"]}
    out = {"dlc": [], "gar_bench": []}
    for a in d["annotations"][:6]:
        seg = ast.literal_eval(a["segmentation"]) if isinstance(a["segmentation"], str) else a["segmentation"]
        bb = ast.literal_eval(a["bbox"]) if isinstance(a["bbox"], str) else a["bbox"]
        im = imgs[str(a["image_id"])]
        out["dlc"].append({"segmentation": seg, "bbox": bb, "image_hw": [im["height"], im["width"]]})
    g = json.load(open(os.path.join(ref, "evaluation/GAR-Bench/annotations/GAR-Bench-VQA.json")))
    for it in g[:3]:
        out["gar_bench"].append({"mask_rles": it["mask_rles"]})
    json.dump(out, open(os.path.join(OUT, "rle_samples.json"), "w"))


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    golden_rle_samples()
    golden_llama()
    golden_llama_padded()
    golden_projector()
    golden_torch_ops()
    golden_ref_helpers()
