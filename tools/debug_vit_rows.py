"""debug: two IDENTICAL tiles through one ViT block — which buffer is the first to differ between tile 0 and tile 1?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from gar_amd import GARConfig, hip, ops
from gar_amd.modeling_gar import GARModel, LOG2E
cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 1})
m = GARModel.from_synthetic(cfg, 0, torch.bfloat16)
v = cfg.mllm_config.vision_config
g = torch.Generator().manual_seed(1)
tile = (torch.rand(1, 3, 448, 448, generator=g) * 2 - 1).to(torch.bfloat16)
mask = torch.full((1, 3, 448, 448), (1 - 127.5) / 127.5).to(torch.bfloat16)
for T in (18, 34):
    pix = tile.repeat(T, 1, 1, 1).cuda()
    msk = mask.repeat(T, 1, 1, 1).cuda()
    p2 = m.get_image_features(pix, msk, pooled=False)
    ws = m._ws[("vit",)]
    N = v.num_patches + m.npt
    def rows(name, shape):
        return ws[name].view(-1)[:torch.Size(shape).numel()].view(*shape)
    D, H, hd, Dm = v.embed_dim, v.num_heads, m.v_hd, v.mlp_dim
    C_l = cfg.mllm_config.text_config.hidden_size
    checks = [("im2col", rows("im2col", (T, v.num_patches, m.Kp))), ("hbuf(LN2 out)", rows("h", (T, N, D))),
              ("Q", rows("Q", (T, H, (N + 63) // 64 * 64, hd))), ("K", rows("K", (T, H, (N + 63) // 64 * 64, hd))),
              ("Vt", rows("Vt", (T, H, hd, (N + 63) // 64 * 64))), ("att", rows("att", (T, N, H * hd))),
              ("x(final)", rows("x", (T, N, D))), ("p2(projector)", p2.view(T, N, C_l))]
    print(f"--- {T} identical tiles")
    for name, t in checks:
        for j in (1, 2, 17, T - 1):
            d = (t[0].float() - t[j].float()).abs()
            nz = int((d > 0).sum())
            msg = "SAME" if nz == 0 else f"DIFF n={nz} max={float(d.max()):.4g} first idx={tuple(int(i) for i in (d > 0).nonzero()[0])}"
            print(f"  {name:16s} tile0 vs tile{j}: {msg}")
