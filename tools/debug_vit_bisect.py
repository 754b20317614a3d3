"""debug: two identical samples of 17 distinct tiles — run the first ViT ops one by one on separate buffers and report the
first op whose output differs between the two halves (and whether a second run reproduces the same values)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from gar_amd import GARConfig, hip, ops
from gar_amd.modeling_gar import GARModel, LOG2E
cfg = GARConfig.gar_1b(**{"vision.depth": 1, "text.num_hidden_layers": 1})
m = GARModel.from_synthetic(cfg, 0, torch.bfloat16, keep_plain_weights=True)   # this tool reads the un-folded n1 / qkv_w / qkv_b
v = cfg.mllm_config.vision_config
dev, dt = "cuda:0", torch.bfloat16
g2 = torch.Generator().manual_seed(2)
half = (torch.rand(17, 3, 448, 448, generator=g2) * 2 - 1).to(dt)
mh = torch.full((17, 3, 448, 448), (1 - 127.5) / 127.5).to(dt)
mh[:, :, 100:200, 50:300] = (5 - 127.5) / 127.5
pix, msk = torch.cat([half, half]).to(dev), torch.cat([mh, mh]).to(dev)
T = 34
n, D, H, hd = v.num_patches, v.embed_dim, v.num_heads, m.v_hd
N, Da = n + m.npt, H * hd
Npad = (N + 63) // 64 * 64
def cmp(name, t, dim0=T):
    a = t.view(dim0, -1)
    d = (a[:dim0 // 2].float() - a[dim0 // 2:].float()).abs()
    nz = int((d > 0).sum())
    rows = sorted(set(int(i) for i in (d > 0).nonzero()[:, 0].tolist()))[:8] if nz else []
    print(f"  {name:28s}: {'SAME' if nz == 0 else f'DIFF n={nz} max={float(d.max()):.4g} tiles={rows}'}", flush=True)
    return nz
for rep in range(2):
    print(f"=== run {rep}")
    A = torch.empty(T * n, m.Kp, dtype=dt, device=dev)
    ops.patch_im2col(pix, msk, A, v.patch_size, cfg.prompt_numbers)
    cmp("im2col", A)
    x = torch.zeros(T, N, D, dtype=dt, device=dev)
    x2 = x.view(T * N, D)
    ops.gemm(A, m.w_patch, x2, hip.EPI_PATCH_POS, pos=m.pos, tokens_in=n, tokens_out=N, token_offset=m.npt)
    cmp("patch GEMM (PATCH_POS)", x)
    ops.cls_pos_fill(x, m.cls, m.pos)
    cmp("+cls", x)
    xa = x.clone()
    ops.layernorm(x2, *m.norm_pre, v.ln_eps)
    cmp("norm_pre (in place)", x)
    xb = torch.empty_like(xa)
    ops.layernorm(xa.view(T * N, D), *m.norm_pre, v.ln_eps, out=xb.view(T * N, D))
    print("    in-place == out-of-place:", torch.equal(x, xb))
    blk = m.vblocks[0]
    hbuf = torch.empty(T * N, D, dtype=dt, device=dev)
    ops.layernorm(x2, *blk["n1"], v.ln_eps, out=hbuf)
    cmp("LN1", hbuf.view(T, N, D))
    qkv = torch.empty(T * N, 3 * Da, dtype=dt, device=dev)
    ops.gemm(hbuf, blk["qkv_w"], qkv, hip.EPI_BIAS, bias=blk["qkv_b"])
    cmp("qkv GEMM (BIAS, unfused)", qkv.view(T, N, 3 * Da))
    qkv2 = torch.empty_like(qkv)
    ops.gemm(hbuf, blk["qkv_w"], qkv2, hip.EPI_BIAS, bias=blk["qkv_b"])
    print("    qkv GEMM run twice identical:", torch.equal(qkv, qkv2))
    Q = torch.zeros(T, H, Npad, hd, dtype=dt, device=dev)
    K = torch.zeros(T, H, Npad, hd, dtype=dt, device=dev)
    vrow = torch.empty(T * N, Da, dtype=dt, device=dev)
    ok = ops.gemm_qkv_rope(hbuf, blk["qkv_w"], blk["qkv_b"], vrow, Q, K, m.vit_sin, m.vit_cos, H, hd, N, Npad, m.npt,
                           (v.head_dim ** -0.5) * LOG2E)
    print("    fused qkv taken:", ok)
    cmp("fused Q", Q); cmp("fused K", K); cmp("fused V rows", vrow.view(T, N, Da))
    Vt = torch.zeros(T, H, hd, Npad, dtype=dt, device=dev)
    ops.vit_v_transpose(vrow, Vt, T, N, H, hd, Npad)
    cmp("Vt", Vt)
    att = torch.empty(T * N, Da, dtype=dt, device=dev)
    ops.attention(Q, K, Vt, att, T, H, H, hd, N, Npad, N, Npad, causal=False)
    cmp("attention", att.view(T, N, Da))
    att2 = torch.empty_like(att)
    ops.attention(Q, K, Vt, att2, T, H, H, hd, N, Npad, N, Npad, causal=False)
    print("    attention run twice identical:", torch.equal(att, att2))
    x3 = x2.clone()
    ops.gemm(att, blk["proj_w"], x3, hip.EPI_BIAS_SCALE_RES, bias=blk["proj_b"], residual=x3, gamma=blk["g1"])
    cmp("proj GEMM (+res, in place)", x3.view(T, N, D))
