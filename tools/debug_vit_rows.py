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
T = 34
g2 = torch.Generator().manual_seed(2)
half = (torch.rand(17, 3, 448, 448, generator=g2) * 2 - 1).to(torch.bfloat16)
mhalf = mask.repeat(17, 1, 1, 1).clone()
mhalf[:, :, 100:200, 50:300] = (5 - 127.5) / 127.5
pix = torch.cat([half, half]).cuda()
msk = torch.cat([mhalf, mhalf]).cuda()
p2 = m.get_image_features(pix, msk, pooled=False)
ws = m._ws[("vit",)]
N = v.num_patches + m.npt
def rows(name, shape):
    return ws[name].view(-1)[:torch.Size(shape).numel()].view(*shape)
D, H, hd, Dm = v.embed_dim, v.num_heads, m.v_hd, v.mlp_dim
C_l = cfg.mllm_config.text_config.hidden_size
Np = (N + 63) // 64 * 64
checks = [("im2col", rows("im2col", (T, v.num_patches, m.Kp))), ("hbuf(LN2 out)", rows("h", (T, N, D))),
          ("Q", rows("Q", (T, H, Np, hd))), ("K", rows("K", (T, H, Np, hd))), ("Vt", rows("Vt", (T, H, hd, Np))),
          ("att", rows("att", (T, N, H * hd))), ("x(final)", rows("x", (T, N, D))), ("p2(projector)", p2.view(T, N, C_l))]
print("--- two identical samples of 17 DISTINCT tiles, one ViT block")
for name, t in checks:
    d = (t[:17].float() - t[17:].float()).abs()
    nz = int((d > 0).sum())
    msg = "SAME" if nz == 0 else f"DIFF n={nz} of {d.numel()} max={float(d.max()):.4g} first idx={tuple(int(i) for i in (d > 0).nonzero()[0])} last idx={tuple(int(i) for i in (d > 0).nonzero()[-1])}"
    print(f"  {name:16s} sample0 vs sample1: {msg}")
