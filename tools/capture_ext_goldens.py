#!/usr/bin/env python
"""One-command pin for the two third-party kernels the oracle restates from their published source (SURVEY.md section 8c,
DESIGN.md section 2 "parity unpinned"):

    python tools/capture_ext_goldens.py            # writes tests/golden/ext_timm_eva.npz and / or ext_tv_roi_align.npz

  (i)  timm (requirements.txt pins 1.0.19): the PE ViT the reference instantiates through AutoModel.from_config ->
       timm `Eva` ("vit_pe_lang_*"; call sites /root/reference/projects/grasp_any_region/models/modeling/
       modeling_perception_lm.py:179,194-216): seeded weights at reduced dims, the RotaryEmbeddingCat tables, every block's
       output and `forward_features` with the reference's custom order (patch_embed -> _pos_embed -> norm_pre -> blocks -> norm);
  (ii) torchvision.ops.roi_align as the reference calls it (/root/reference/projects/grasp_any_region/hf_models/
       modeling_gar.py:389-396: fp32 map, output 16 x 16, spatial_scale 1/28, sampling_ratio 2, aligned=True): the demo-1 box,
       edge boxes (clamps, out-of-map samples, last-cell branch), a box straddling tiles, and the video path's 16 x 16 map.

Neither package is installable in the build image (no network; probed every round), so this script has never run there: the day
a box has `timm` / `torchvision`, running it and committing the two .npz files turns tests/test_oracle_goldens.py's
`test_ext_*` from SKIPPED into the pin — no harness left to write. The fixtures are data only (inputs, seeded weights, outputs)."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")

# reduced PE-Lang-shaped tower: 4 x 4 grid of 14-px patches, 2 blocks, cls token (L/14's structure), head_dim 32
EVA = dict(img_size=56, patch_size=14, embed_dim=64, depth=2, num_heads=2, mlp_ratio=4.0, ref_feat_shape=(4, 4))
# reduced PE-G-shaped tower (GAR-8B): NO cls token, head_dim 48 (not a power of two, like G/14's 96)
EVA_G = dict(img_size=56, patch_size=14, embed_dim=96, depth=2, num_heads=2, mlp_ratio=4.0, ref_feat_shape=(4, 4))


def capture_timm(out_path):
    import timm
    made = {}
    for tag, variant, dims in (("l", "vit_pe_lang_large_patch14_448", EVA), ("g", "vit_pe_lang_gigantic_patch14_448", EVA_G)):
        torch.manual_seed(1234)
        # the release variant's own model_args carry the PE flags (rope, abs pos-embed, LayerScale init, norm placement, cls
        # token or not): only the sizes are overridden
        model = timm.create_model(variant, pretrained=False, **dims)
        model.eval()
        with torch.no_grad():
            for n, p in model.named_parameters():              # seeded, non-degenerate (LayerScale at its init value 0.1)
                if n.endswith(("gamma_1", "gamma_2")):
                    p.fill_(0.1)
                elif p.dim() == 1 and ("norm" in n and n.endswith("weight")):
                    p.copy_(1.0 + 0.1 * torch.randn_like(p))
                else:
                    p.copy_(0.5 * torch.randn_like(p) * (p.shape[-1] ** -0.5 if p.dim() > 1 else 0.2))
            x = torch.randn(3, 3, dims["img_size"], dims["img_size"])
            mask_embeds = 0.3 * torch.randn(3, dims["embed_dim"], 4, 4)
            # the reference's custom_forward_features, statement by statement (modeling_perception_lm.py:194-216)
            h = model.patch_embed(x)
            h = h + mask_embeds.flatten(2).transpose(1, 2)
            h, rot = model._pos_embed(h)
            h = model.norm_pre(h)
            blocks = []
            for blk in model.blocks:
                h = blk(h, rope=rot)
                blocks.append(h.clone())
            h = model.norm(h)
            made[f"{tag}_input"] = x.numpy()
            made[f"{tag}_mask_embeds"] = mask_embeds.numpy()
            made[f"{tag}_rope"] = rot.numpy()                  # RotaryEmbeddingCat.get_embed(): cat(sin, cos) [n, 2 hd]
            for i, b in enumerate(blocks):
                made[f"{tag}_block{i}"] = b.numpy()
            made[f"{tag}_out"] = h.numpy()
            made[f"{tag}_plain_forward_features"] = model.forward_features(x).numpy()       # timm's own order, no mask_embeds
            for k, v in model.state_dict().items():
                made[f"{tag}_w/{k}"] = v.numpy()
            made[f"{tag}_dims"] = np.array([dims["img_size"], dims["patch_size"], dims["embed_dim"], dims["depth"],
                                            dims["num_heads"], int(dims["embed_dim"] * dims["mlp_ratio"]),
                                            int(getattr(model, "num_prefix_tokens", 0))], dtype=np.int64)
    made["timm_version"] = np.array(timm.__version__)
    np.savez_compressed(out_path, **made)
    print(f"wrote {out_path} (timm {timm.__version__}; requirements.txt pins 1.0.19)")


def capture_torchvision(out_path):
    import torchvision
    from torchvision.ops import roi_align
    g = torch.Generator().manual_seed(77)
    made = {}
    cases = {
        # (map [1, C, H, W], boxes [K, 5] in the reference's `roi_feat` coordinates, spatial_scale)
        "demo1": (torch.randn(1, 8, 64, 64, generator=g),
                  [[0, 0.720703125 * 64 * 28 * (1 / 28), 0.8688311688311688 * 64 * 28 * (1 / 28),
                    0.7939453125 * 64 * 28 * (1 / 28), 0.9233766233766234 * 64 * 28 * (1 / 28)]], 1 / 28),
        "edges": (torch.randn(1, 5, 32, 48, generator=g),
                  [[0, -40.0, -30.0, 200.0, 100.0], [0, 1300.0, 850.0, 1344.0, 896.0], [0, 0.0, 0.0, 1344.0, 896.0],
                   [0, 600.0, 400.0, 600.5, 400.5], [0, 430.0, 430.0, 470.0, 470.0]], 1 / 28),
        "video16": (torch.randn(1, 6, 16, 16, generator=g), [[0, 3.0, 2.0, 11.5, 14.0], [0, 0.0, 0.0, 16.0, 16.0]], 1 / 28),
        "unaligned_scale1": (torch.randn(1, 4, 20, 20, generator=g), [[0, 4.0, 4.0, 20.0, 16.0]], 1.0),
    }
    for name, (fm, boxes, scale) in cases.items():
        rois = torch.tensor(boxes, dtype=torch.float32)
        out = roi_align(fm.float(), rois, output_size=(16, 16), spatial_scale=scale, sampling_ratio=2, aligned=True)
        made[f"{name}_map"], made[f"{name}_rois"], made[f"{name}_scale"] = fm.numpy(), rois.numpy(), np.float32(scale)
        made[f"{name}_out"] = out.numpy()
        if name == "unaligned_scale1":
            made[f"{name}_out_aligned_false"] = roi_align(fm.float(), rois, output_size=(16, 16), spatial_scale=scale,
                                                          sampling_ratio=2, aligned=False).numpy()
    made["torchvision_version"] = np.array(torchvision.__version__)
    np.savez_compressed(out_path, **made)
    print(f"wrote {out_path} (torchvision {torchvision.__version__})")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out-dir", default=GOLD)
    a = ap.parse_args()
    done = 0
    for name, fn, mod in (("ext_timm_eva.npz", capture_timm, "timm"), ("ext_tv_roi_align.npz", capture_torchvision, "torchvision")):
        try:
            __import__(mod)
        except Exception as e:                                   # noqa: BLE001 — report and go on to the other package
            print(f"{mod} is not importable here ({type(e).__name__}: {e}): {name} not captured")
            continue
        fn(os.path.join(a.out_dir, name))
        done += 1
    if not done:
        print("nothing captured: this box has neither package (the build image does not; see the module docstring)")
        return 1
    print("now run: python -m pytest tests/test_oracle_goldens.py -k ext -q   and commit the .npz files")
    return 0


if __name__ == "__main__":
    sys.exit(main())
