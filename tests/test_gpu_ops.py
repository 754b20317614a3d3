"""GPU (-m gpu): every HIP kernel behind the C ABI against a plain fp32/fp64 PyTorch-CPU statement of the same op
or against the oracle, on seeded inputs. f32 mode = tight tolerance (exact-f32 MFMA, only summation order differs);
bf16 mode = bf16-rounding tolerance; index / selection work is bit-exact."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# float16: the twin library libgar_hip_f16.so (the same kernels with IEEE binary16 as the 16-bit type, hip.lib(dtype))
DT = [torch.float32, torch.bfloat16, torch.float16]
HALF = [torch.bfloat16, torch.float16]


def tol(dt):
    return {torch.float32: 2e-5, torch.bfloat16: 1.6e-2, torch.float16: 2e-3}[dt]


def close(out, ref, dt, scale=None, extra=1.0):
    out = out.detach().float().cpu().double()
    ref = ref.detach().double().cpu()
    s = float(ref.abs().max()) if scale is None else scale
    err = float((out - ref).abs().max())
    assert err <= extra * tol(dt) * max(s, 1e-6) + 1e-6, f"max err {err} vs scale {s} ({dt})"


@pytest.fixture(scope="module")
def dev():
    from gar_amd import hip
    hip.require_device(0)
    return torch.device("cuda:0")


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def q(x, dt):
    """value as the kernel sees it (bf16-rounded inputs are the reference's inputs too)"""
    return x.to(dt).float()


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (300, 256, 192), (1025, 384, 128), (17, 1000, 64), (5, 48, 256),
                                   (16, 262, 128), (1, 64, 2048)])
def test_gemm_plain_and_tails(dev, dt, M, N, K):
    from gar_amd import hip, ops
    if N % 16 and M > 16 and N % 4:
        pytest.skip("general kernel needs N%4==0")
    a, w = q(rnd(M, K, seed=1), dt), q(rnd(N, K, seed=2, scale=K ** -0.5), dt)
    ldc = (N + 63) // 64 * 64
    out = torch.full((M, ldc), 7.0, dtype=dt, device=dev)
    ops.gemm(a.to(dev, dt), w.to(dev, dt), out)
    ref = a.double() @ w.double().T
    close(out[:, :N], ref, dt)
    assert float((out[:, N:].float() - 7.0).abs().max()) == 0 if ldc > N else True


@pytest.mark.parametrize("M,N,K", [(4099, 2048, 192), (8192, 1024, 64), (2049, 4096, 1024), (5000, 1968, 128),
                                   (16640, 2048, 64), (9000, 2048, 128), (33000, 1024, 320)])
@pytest.mark.parametrize("dt", HALF)
def test_gemm_bf16_pingpong_kernel(dev, M, N, K, dt):
    """shapes large enough (>= 128 tiles of 256x256) to take the ping-pong kernel, incl. M / N tails, every epilogue,
    one / two / odd numbers of K tiles and more tiles than CUs (the persistent loop's prefetch wraps into the next
    output tile at every position of the rotated K loop); compared element-wise with an fp64 reference of the same
    bf16 inputs."""
    from gar_amd import hip, ops
    a, w = q(rnd(M, K, seed=50), dt), q(rnd(N, K, seed=51, scale=K ** -0.5), dt)
    A, W_ = a.to(dev, dt), w.to(dev, dt)
    acc = (a.to(dev).double() @ w.to(dev).double().T).cpu()
    out = torch.full((M, N), 9.0, dtype=dt, device=dev)
    ops.gemm(A, W_, out)
    close(out, acc, dt)
    bias, gamma, res = q(rnd(N, seed=52), dt), q(rnd(N, seed=53), dt), q(rnd(M, N, seed=54), dt)
    ops.gemm(A, W_, out, hip.EPI_BIAS_GELU, bias=bias.to(dev, dt))
    close(out, F.gelu(acc + bias.double()), dt)
    # the GELU table's ends: inputs far outside [-8, 8) (|x| up to ~40) must come out as x resp. 0, to bf16 rounding of
    # the VALUE (not of the table's range) — the table extends its last / first interval instead of testing the range
    wide = q(rnd(N, seed=55) * 12.0, dt)
    ops.gemm(A, W_, out, hip.EPI_BIAS_GELU, bias=wide.to(dev, dt))
    refw = F.gelu(acc + wide.double())
    errw = (out.float().cpu().double() - refw).abs()
    assert float((errw - 2.0 ** -8 * refw.abs()).max()) <= 1e-5, float(errw.max())
    ops.gemm(A, W_, out, hip.EPI_BIAS, bias=bias.to(dev, dt))
    close(out, acc + bias.double(), dt)
    r = res.to(dev, dt).clone()
    ops.gemm(A, W_, r, hip.EPI_RES, residual=r)
    close(r, res.double() + acc, dt)
    r = res.to(dev, dt).clone()
    ops.gemm(A, W_, r, hip.EPI_BIAS_SCALE_RES, bias=bias.to(dev, dt), residual=r, gamma=gamma.to(dev, dt))
    close(r, res.double() + gamma.double() * (acc + bias.double()), dt)
    if N % 32 == 0:
        Fd = N // 2
        g_w, u_w = w[:Fd], w[Fd:]
        gu = torch.stack([g_w.view(Fd // 16, 16, K), u_w.view(Fd // 16, 16, K)], 1).reshape(N, K)
        o2 = torch.empty(M, Fd, dtype=dt, device=dev)
        ops.gemm(A, gu.to(dev, dt), o2, hip.EPI_SWIGLU)
        close(o2, F.silu(acc[:, :Fd]) * acc[:, Fd:], dt)


@pytest.mark.parametrize("dt", HALF)
def test_gemm_bf16_pingpong_patch_pos_epilogue(dev, dt):
    """GAR_EPI_PATCH_POS at a size that takes the ping-pong kernel (row m -> token 1 + m % n of tile m / n of a
    [tiles, n + 1, N] output, + pos-embed row): element-wise against fp64; the cls slots stay untouched."""
    from gar_amd import hip, ops
    T, n, N, K = 16, 1024, 1024, 128
    a, w = q(rnd(T * n, K, seed=90), dt), q(rnd(N, K, seed=91, scale=K ** -0.5), dt)
    pos = q(rnd(n + 1, N, seed=92, scale=0.2), dt)
    x = torch.full((T, n + 1, N), 3.0, dtype=dt, device=dev)
    ops.gemm(a.to(dev, dt), w.to(dev, dt), x.view(T * (n + 1), N), hip.EPI_PATCH_POS, pos=pos.to(dev, dt), tokens_in=n,
             tokens_out=n + 1, token_offset=1)
    ref = (a.double() @ w.double().T).view(T, n, N) + pos[1:].double()
    close(x[:, 1:], ref, dt)
    assert float((x[:, 0].float() - 3.0).abs().max()) == 0.0


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("M", [4, 200])
def test_gemm_epilogues(dev, dt, M):
    from gar_amd import hip, ops
    N, K = 256, 128
    a, w = q(rnd(M, K, seed=3), dt), q(rnd(N, K, seed=4, scale=K ** -0.5), dt)
    bias, gamma, res = q(rnd(N, seed=5), dt), q(rnd(N, seed=6), dt), q(rnd(M, N, seed=7), dt)
    A, W_ = a.to(dev, dt), w.to(dev, dt)
    acc = a.double() @ w.double().T
    out = torch.empty(M, N, dtype=dt, device=dev)
    ops.gemm(A, W_, out, hip.EPI_BIAS, bias=bias.to(dev, dt))
    close(out, acc + bias.double(), dt)
    ops.gemm(A, W_, out, hip.EPI_BIAS_GELU, bias=bias.to(dev, dt))
    close(out, F.gelu(acc + bias.double()), dt)
    ops.gemm(A, W_, out, hip.EPI_BIAS, bias=bias.to(dev, dt))
    close(out, acc + bias.double(), dt)
    r = res.to(dev, dt).clone()
    ops.gemm(A, W_, r, hip.EPI_RES, residual=r)
    close(r, res.double() + acc, dt)
    r = res.to(dev, dt).clone()
    ops.gemm(A, W_, r, hip.EPI_BIAS_SCALE_RES, bias=bias.to(dev, dt), residual=r, gamma=gamma.to(dev, dt))   # in place
    close(r, res.double() + gamma.double() * (acc + bias.double()), dt)
    r = res.to(dev, dt).clone()
    ops.gemm(A, W_, r, hip.EPI_RES, residual=r)
    close(r, res.double() + acc, dt)
    # SWIGLU with the [gate16|up16] interleave
    Fd = N // 2
    g_w, u_w = w[:Fd], w[Fd:]
    gu = torch.stack([g_w.view(Fd // 16, 16, K), u_w.view(Fd // 16, 16, K)], 1).reshape(N, K)
    o2 = torch.empty(M, Fd, dtype=dt, device=dev)
    ops.gemm(A, gu.to(dev, dt), o2, hip.EPI_SWIGLU)
    close(o2, F.silu(a.double() @ g_w.double().T) * (a.double() @ u_w.double().T), dt)


@pytest.mark.parametrize("dt", DT)
def test_gemm_strided_rows(dev, dt):
    from gar_amd import ops
    B, S, Cc, N = 3, 10, 128, 64
    h = q(rnd(B, S, Cc, seed=8), dt)
    w = q(rnd(N, Cc, seed=9, scale=Cc ** -0.5), dt)
    hd = h.to(dev, dt)
    out = torch.empty(B, N, dtype=dt, device=dev)
    ops.gemm(hd[:, S - 1, :], w.to(dev, dt), out)
    close(out, h[:, S - 1].double() @ w.double().T, dt)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("D", [128, 1024, 2048, 4608])
def test_norms(dev, dt, D):
    from gar_amd import ops
    M = 37
    x = q(rnd(M, D, seed=10, scale=2.0) + 0.3, dt)
    w, b = q(1 + 0.1 * rnd(D, seed=11), dt), q(0.1 * rnd(D, seed=12), dt)
    y = torch.empty(M, D, dtype=dt, device=dev)
    ops.layernorm(x.to(dev, dt), w.to(dev, dt), b.to(dev, dt), 1e-5, out=y)
    close(y, F.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5), dt)
    ops.rmsnorm(x.to(dev, dt), w.to(dev, dt), 1e-5, out=y)
    xd = x.double()
    n = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5)
    if dt in HALF:
        n = n.to(dt).double()
    close(y, w.double() * n, dt)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("D", [192, 1024])
def test_norms_two_rows_per_wave_path(dev, D, dt):
    """bf16, D <= 1024, M >= 4096 takes the two-rows-per-wave kernel: odd M (last wave has one row), in-place, and it
    must agree bit for bit with the one-row kernel (same arithmetic order per row)."""
    from gar_amd import ops
    M = 4097
    x = q(rnd(M, D, seed=13, scale=2.0) + 0.3, dt)
    w, b = q(1 + 0.1 * rnd(D, seed=14), dt), q(0.1 * rnd(D, seed=15), dt)
    xd_, wd_, bd_ = x.to(dev, dt), w.to(dev, dt), b.to(dev, dt)
    y = torch.empty(M, D, dtype=dt, device=dev)
    ops.layernorm(xd_, wd_, bd_, 1e-5, out=y)
    close(y, F.layer_norm(x.double(), (D,), w.double(), b.double(), 1e-5), dt)
    y1 = torch.empty(37, D, dtype=dt, device=dev)
    ops.layernorm(xd_[-37:].contiguous(), wd_, bd_, 1e-5, out=y1)          # M = 37: one-row kernel
    assert torch.equal(y[-37:], y1)
    z = xd_.clone()
    ops.layernorm(z, wd_, bd_, 1e-5)                                        # in place
    assert torch.equal(z, y)
    # a row's result must not depend on whether it is the first or the second row of its wave (= on the position of its
    # sample in a batch): the same rows shifted by one change parity and must come out bit-identical
    ys = torch.empty(M - 1, D, dtype=dt, device=dev)
    ops.layernorm(xd_[1:].contiguous(), wd_, bd_, 1e-5, out=ys)
    assert torch.equal(ys, y[1:])
    rs, r0 = torch.empty(M - 1, D, dtype=dt, device=dev), torch.empty(M, D, dtype=dt, device=dev)
    ops.rmsnorm(xd_, wd_, 1e-5, out=r0)
    ops.rmsnorm(xd_[1:].contiguous(), wd_, 1e-5, out=rs)
    assert torch.equal(rs, r0[1:])
    ops.rmsnorm(xd_, wd_, 1e-5, out=y)
    ops.rmsnorm(xd_[-37:].contiguous(), wd_, 1e-5, out=y1)
    assert torch.equal(y[-37:], y1)


# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dt", DT)
def test_patch_embed_with_mask_matches_two_convs(dev, dt):
    """im2col + one GEMM (PATCH_POS epilogue) == patch_embed conv + mask conv + pos embed (oracle A1/A2/A4 head)."""
    from gar_amd import hip, ops
    from oracle import gar_oracle as O
    T, img, patch, D, P = 3, 56, 14, 64, 5
    g = img // patch
    n = g * g
    pix = q(rnd(T, 3, img, img, seed=13), dt)
    ids = torch.randint(0, 8, (T, 1, img, img), generator=torch.Generator().manual_seed(14)).expand(T, 3, img, img)
    mvals = q((ids.float() / 255.0 - 0.5) / 0.5, dt)
    wp, wm = q(rnd(D, 3, patch, patch, seed=15, scale=0.05), dt), q(rnd(D, 3, patch, patch, seed=16, scale=0.05), dt)
    pos = q(rnd(n + 1, D, seed=17, scale=0.2), dt)
    Kp = (6 * patch * patch + 63) // 64 * 64
    wcat = torch.zeros(D, Kp)
    wcat[:, :3 * patch * patch] = wp.flatten(1)
    wcat[:, 3 * patch * patch:6 * patch * patch] = wm.flatten(1)
    A = torch.empty(T * n, Kp, dtype=dt, device=dev)
    ops.patch_im2col(pix.to(dev, dt), mvals.to(dev, dt), A, patch, P)
    x = torch.zeros(T, n + 1, D, dtype=dt, device=dev)
    ops.gemm(A, wcat.to(dev, dt), x.view(T * (n + 1), D), hip.EPI_PATCH_POS, pos=pos.to(dev, dt), tokens_in=n,
             tokens_out=n + 1, token_offset=1)
    binary = O.decode_mask_values(mvals.to(dt), P)
    assert binary.unique().tolist() == [0.0, 1.0]
    ref = F.conv2d(pix.double(), wp.double(), stride=patch) + F.conv2d(binary.double(), wm.double(), stride=patch)
    ref = ref.flatten(2).transpose(1, 2) + pos[1:].double()
    close(x[:, 1:], ref, dt)
    assert float(x[:, 0].float().abs().max()) == 0.0
    # binary mask columns are exact
    Ah = A.float().cpu().view(T, g, g, Kp)[..., 3 * patch * patch:6 * patch * patch].reshape(T, g, g, 3, patch, patch)
    bref = binary.view(T, 3, g, patch, g, patch).permute(0, 2, 4, 1, 3, 5)
    assert torch.equal(Ah, bref)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("T,npt,with_mask", [(9, 1, True), (8, 0, True), (9, 1, False)])
def test_patch_embed_gather_matches_two_convs(dev, T, npt, with_mask, dt):
    """gar_mask_decode + gar_patch_embed (patches DMA'd from the image tiles into LDS by the tile GEMM, weights in the
    gather's K order) == patch-embed conv + mask conv + pos embed in fp64, and == the im2col + GEMM path within bf16
    rounding; reads that stray out of the tensor with a non-zero weight would show (neighbour images hold 1e4)."""
    from gar_amd import hip, ops
    from oracle import gar_oracle as O
    img, patch, D, P = 448, 14, 1024, 5
    g = img // patch
    n = g * g
    Kg = ops.patch_embed_k(img, patch)
    assert Kg == 1536
    pix = q(rnd(T, 3, img, img, seed=23), dt)
    ids = torch.randint(0, 8, (T, 3, img, img), generator=torch.Generator().manual_seed(24))      # channels differ
    mvals = q((ids.float() / 255.0 - 0.5) / 0.5, dt)
    wp, wm = q(rnd(D, 3, patch, patch, seed=25, scale=0.05), dt), q(rnd(D, 3, patch, patch, seed=26, scale=0.05), dt)
    pos = q(rnd(n + npt, D, seed=27, scale=0.2), dt)
    wg = torch.zeros(D, 2, 3, 16, 16)
    wg[:, 0, :, :patch, :patch] = wp
    wg[:, 1, :, :patch, :patch] = wm
    wg = wg.view(D, 2, 3, 4, 4, 16).reshape(D, Kg)
    big = torch.full((T + 2, 3, img, img), 1e4, dtype=dt, device=dev)
    big[1:T + 1] = pix.to(dev, dt)
    pixd = big[1:T + 1]
    binary = O.decode_mask_values(mvals.to(dt), P) if with_mask else torch.zeros_like(mvals)
    mb = torch.empty(T, 3, img, img, dtype=dt, device=dev)
    if with_mask:
        ops.mask_decode(mvals.to(dev, dt), mb, P)
        assert torch.equal(mb.cpu().float(), binary.float())                  # A1 alone: bit-exact
    else:
        mb.zero_()
    x = torch.zeros(T, n + npt, D, dtype=dt, device=dev)
    assert ops.patch_embed(pixd, mb, wg.to(dev, dt), pos.to(dev, dt), x, patch, npt)
    torch.cuda.synchronize()
    ref = F.conv2d(pix.double(), wp.double(), stride=patch) + F.conv2d(binary.double(), wm.double(), stride=patch)
    ref = ref.flatten(2).transpose(1, 2) + pos[npt:].double()
    close(x[:, npt:], ref, dt)
    if npt:
        assert float(x[:, 0].float().abs().max()) == 0.0                      # cls rows untouched
    # the im2col form of the same product (different K order: fp32 sums may differ in the last bits before rounding)
    Kp = (6 * patch * patch + 63) // 64 * 64
    wcat = torch.zeros(D, Kp)
    wcat[:, :3 * patch * patch] = wp.flatten(1)
    wcat[:, 3 * patch * patch:6 * patch * patch] = wm.flatten(1)
    A = torch.empty(T * n, Kp, dtype=dt, device=dev)
    ops.patch_im2col(pixd.contiguous(), mvals.to(dev, dt) if with_mask else None, A, patch, P)
    x2 = torch.zeros(T, n + npt, D, dtype=dt, device=dev)
    ops.gemm(A, wcat.to(dev, dt), x2.view(T * (n + npt), D), hip.EPI_PATCH_POS, pos=pos.to(dev, dt), tokens_in=n,
             tokens_out=n + npt, token_offset=npt)
    d = (x.float() - x2.float()).abs()
    assert float(d.max()) <= 2.0 ** -7 * float(x2.float().abs().max()), float(d.max())
    assert float((d > 0).float().mean()) < 0.05                               # rare last-bit flips only


@pytest.mark.parametrize("dt", HALF)
def test_patch_embed_gather_refuses_other_shapes(dev, dt):
    from gar_amd import ops
    for T, img, patch, D in [(3, 56, 14, 64), (2, 448, 14, 512), (9, 512, 16, 1024)]:      # grid 4; < 128 tiles; taken
        pix = torch.zeros(T, 3, img, img, dtype=dt, device=dev)
        n = (img // patch) ** 2
        x = torch.zeros(T, n, D, dtype=dt, device=dev)
        Kg = ops.patch_embed_k(img, patch)
        took = ops.patch_embed(pix, pix, torch.zeros(D, Kg, dtype=dt, device=dev), torch.zeros(n, D, dtype=dt, device=dev),
                               x, patch, 0)
        assert took == (img == 512)


@pytest.mark.parametrize("dt", DT)
def test_pool2x2_with_cls_window(dev, dt):
    from gar_amd import ops
    T, g, Cc = 2, 8, 64
    x = q(rnd(T, g * g + 1, Cc, seed=18), dt)
    y = torch.empty(T, (g // 2) ** 2, Cc, dtype=dt, device=dev)
    ops.pool2x2(x.to(dev, dt), y, g, in_tile_tokens=g * g + 1, in_token_offset=1)
    ref = F.adaptive_avg_pool2d(x[:, 1:].double().permute(0, 2, 1).reshape(T, Cc, g, g), (g // 2, g // 2))
    close(y, ref.flatten(2).transpose(1, 2), dt)


# ------------------------------------------------------------------------------------------------------------------
def _attn_ref(qh, kh, vh, causal, off):
    s = (qh.double() @ kh.double().transpose(-1, -2)) * (qh.shape[-1] ** -0.5)
    if causal:
        Sq, Sk = s.shape[-2:]
        m = torch.ones(Sq, Sk, dtype=torch.bool).tril(diagonal=off)
        s = s.masked_fill(~m, float("-inf"))
    return torch.softmax(s, -1) @ vh.double()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd", [64, 96, 128])
@pytest.mark.parametrize("N,npt", [(65, 1), (1025, 1), (200, 0)])
def test_vit_attention_path(dev, dt, N, npt, hd):
    """qkv_post (interleaved 2-D RoPE, relayout, Vt) + non-causal attention vs the oracle's AttentionRope math; head_dim
    64 (PE-L), 128, and 96 (PE-G/14: native in bf16 — 192-byte K rows in a 256-byte LDS pitch; the f32 parity mode
    runs it zero-padded to 128 from the model side)."""
    from gar_amd import ops
    from oracle import gar_oracle as O
    if hd == 96 and dt == torch.float32:
        pytest.skip("head_dim 96 is built for bf16 (f32 parity mode pads to 128 in GARModel)")
    T, H = 2, 2
    D = H * hd
    Npad = (N + 63) // 64 * 64
    qkv = q(rnd(T * N, 3 * D, seed=19), dt)
    sin, cos = rnd(N - npt, hd, seed=20).sin(), rnd(N - npt, hd, seed=21).cos()
    Q = torch.full((T, H, Npad, hd), float("nan"), dtype=dt, device=dev)
    K = torch.full((T, H, Npad, hd), float("nan"), dtype=dt, device=dev)
    Vt = torch.full((T, H, hd, Npad), float("nan"), dtype=dt, device=dev)
    out = torch.empty(T * N, D, dtype=dt, device=dev)
    ops.vit_qkv_post(qkv.to(dev, dt), sin.to(dev), cos.to(dev), Q, K, Vt, T, N, npt, H, hd, Npad,
                     (hd ** -0.5) * 1.4426950408889634)
    ops.attention(Q, K, Vt, out, T, H, H, hd, N, Npad, N, Npad, causal=False)
    x = qkv.view(T, N, 3, H, hd).permute(2, 0, 3, 1, 4)
    qq, kk, vv = x.unbind(0)
    rot = O._rot_interleaved
    qq = torch.cat([qq[:, :, :npt], qq[:, :, npt:] * cos + rot(qq[:, :, npt:]) * sin], 2)
    kk = torch.cat([kk[:, :, :npt], kk[:, :, npt:] * cos + rot(kk[:, :, npt:]) * sin], 2)
    ref = _attn_ref(qq, kk, vv, False, 0).transpose(1, 2).reshape(T * N, D)
    close(out, ref, dt, extra=2.0)
    assert torch.isfinite(out.float()).all()
    if dt in HALF:
        # V row-major [T, H, Npad, hd] (what the fused qkv GEMM writes), transposed by the attention's LDS reads
        # (ds_read_b64_tr_b16): same P, same V, same MFMA order -> bit-identical to the Vt form. Pad rows hold large
        # finite values: they meet P = 0 only.
        Vr = torch.full((T, H, Npad, hd), 1e4, dtype=dt, device=dev)
        Vr[:, :, :N] = Vt.transpose(2, 3)[:, :, :N]
        out2 = torch.empty_like(out)
        ops.attention(Q, K, Vr, out2, T, H, H, hd, N, Npad, N, Npad, causal=False, v_row_major=True)
        assert torch.equal(out2, out)
        # kv_prefix = 1: key / value row 0 (the cls token) enters through the initial softmax state (m0 = q.k0, l0 = 1,
        # O0 = v0) and the tiles cover rows 1 ..: 1 + 1024 keys are 16 tiles. Same softmax, another summation order.
        out3 = torch.full_like(out, float("nan"))
        ops.attention(Q, K, Vr, out3, T, H, H, hd, N, Npad, N, Npad, causal=False, v_row_major=True, kv_prefix=1)
        close(out3, ref, dt, extra=2.0)
        close(out3, out.float(), dt, extra=0.5)
        outc, outc2 = torch.empty_like(out), torch.empty_like(out)
        ops.attention(Q, K, Vt, outc, T, H, H, hd, N, Npad, N, Npad, causal=True)
        ops.attention(Q, K, Vr, outc2, T, H, H, hd, N, Npad, N, Npad, causal=True, v_row_major=True)
        assert torch.equal(outc2, outc)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("hd", [64, 96, 128])
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("profile", ["rising", "spike", "falling", "huge"])
def test_attention_lazy_max_redo_path(dev, hd, causal, profile, dt):
    """The bf16 attention exponentiates every tile after the first against the STANDING running max and only checks the
    row sums (attention_bf16.hip, lazy running max); a tile whose scores outgrow that max by more than 2^16 must send the
    wave through the exact path (re-base, redo). Score profiles that force it: keys whose magnitude rises tile by tile,
    one spike tile in the middle, and a fall (the cheap path all the way, with p underflowing to 0); 'huge' jumps by
    more than 2^128 in one tile (exp2 overflows to inf on the lazy pass: the check must still catch it)."""
    from gar_amd import ops
    B, H, n = 1, 2, 64 * 9
    g = torch.Generator().manual_seed(77)
    Qf = torch.randn(B, H, n, hd, generator=g)
    Kf = torch.randn(B, H, n, hd, generator=g)
    Vf = torch.randn(B, H, n, hd, generator=g)
    tile = torch.arange(n) // 64
    scale = {"rising": 1.0 + 8.0 * tile.float(), "spike": torch.where(tile == 4, 70.0, 1.0) * torch.ones(n),
             "falling": 30.0 / (1.0 + 4.0 * tile.float()), "huge": torch.where(tile >= 5, 400.0, 1.0) * torch.ones(n)}[profile]
    # every query has a component 4 along a common unit vector u and the keys of a tile are scale * u (+ noise): the
    # log2-domain scores of a tile are ~ 0.72 * scale * (8 / sqrt(hd)) for every row
    u = torch.nn.functional.normalize(torch.randn(hd, generator=g), dim=0)
    Qf = Qf + 4.0 * u
    Kf = Kf * 0.1 + (hd / 64) ** 0.5 * scale[None, None, :, None] * u
    qs = hd ** -0.5 * 1.4426950408889634
    Q = q(Qf * qs, dt).to(dev, dt)
    K = q(Kf, dt).to(dev, dt)
    Vt = q(Vf, dt).transpose(2, 3).contiguous().to(dev, dt)
    out = torch.empty(B * n, H * hd, dtype=dt, device=dev)
    ops.attention(Q, K, Vt, out, B, H, H, hd, n, n, n, n, causal=causal)
    s = (Q.double().cpu() @ K.double().cpu().transpose(-1, -2)) * 0.6931471805599453       # Q carries scale * log2(e)
    if causal:
        s = s.masked_fill(~torch.ones(n, n, dtype=torch.bool).tril(), float("-inf"))
    ref = (torch.softmax(s, -1) @ Vt.double().cpu().transpose(2, 3)).transpose(1, 2).reshape(B * n, H * hd)
    if profile != "falling" and not causal:
        # the profile does what it is for: past tile 0 the row maxima outgrow the first tile's by more than 2^16
        s2 = s * 1.4426950408889634
        growth = s2[..., 64:].max(-1).values - s2[..., :64].max(-1).values
        assert float(growth.median()) > 20.0 and float((growth > 17.0).float().mean()) > 0.9, float(growth.median())
    assert torch.isfinite(out.float()).all()
    close(out, ref, dt, extra=2.0)
    if True:
        out2 = torch.empty_like(out)
        ops.attention(Q, K, Vt.transpose(2, 3).contiguous(), out2, B, H, H, hd, n, n, n, n, causal=causal, v_row_major=True)
        assert torch.equal(out2, out)
    if not causal:
        # key row 0 through the initial softmax state (kv_prefix = 1): the standing max then starts at q.k0, which the
        # profile outgrows (or undercuts) like any first tile; also with a k0 that dominates / is negligible
        Vr = Vt.transpose(2, 3).contiguous()
        for k0_scale in (1.0, 60.0, -60.0):
            K2 = K.clone()
            K2[:, :, 0] = (K[:, :, 0].float() * k0_scale).to(dt)
            s3 = (Q.double().cpu() @ K2.double().cpu().transpose(-1, -2)) * 0.6931471805599453
            ref3 = (torch.softmax(s3, -1) @ Vr.double().cpu()).transpose(1, 2).reshape(B * n, H * hd)
            out3 = torch.full_like(out, float("nan"))
            ops.attention(Q, K2, Vr, out3, B, H, H, hd, n, n, n, n, causal=False, v_row_major=True, kv_prefix=1)
            assert torch.isfinite(out3.float()).all()
            close(out3, ref3, dt, extra=2.0)


def test_attention_vrow_refuses_what_is_not_built(dev):
    """f32 row-major V: head_dim 64 / 128 only, no kv_prefix (the parity path has no cls-key fold); nothing launched."""
    from gar_amd import hip, ops
    Q = torch.zeros(1, 1, 64, 96, dtype=torch.float32, device=dev)
    with pytest.raises(hip.GarError):
        ops.attention(Q, Q, Q, torch.zeros(64, 96, dtype=torch.float32, device=dev), 1, 1, 1, 96, 64, 64, 64, 64,
                      causal=False, v_row_major=True)
    Q = torch.zeros(1, 1, 64, 64, dtype=torch.float32, device=dev)
    with pytest.raises(hip.GarError):
        ops.attention(Q, Q, Q, torch.zeros(64, 64, dtype=torch.float32, device=dev), 1, 1, 1, 64, 64, 64, 64, 64,
                      causal=False, v_row_major=True, kv_prefix=1)


@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("causal", [False, True])
def test_attention_f32_row_major_v_equals_transposed(dev, hd, causal):
    """parity mode: the f32 kernel stages a row-major V tile transposed — same arithmetic, bit-identical output."""
    from gar_amd import ops
    B, Hq, Hkv, n = 2, 4, 2, 150
    npad = 192
    Q = (rnd(B, Hq, npad, hd, seed=5) * 0.2).to(dev)
    K = rnd(B, Hkv, npad, hd, seed=6).to(dev)
    Vt = rnd(B, Hkv, hd, npad, seed=7).to(dev)
    Vr = Vt.transpose(2, 3).contiguous()
    a = torch.empty(B * n, Hq * hd, device=dev)
    b = torch.empty_like(a)
    ops.attention(Q, K, Vt, a, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal)
    ops.attention(Q, K, Vr, b, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal, v_row_major=True)
    assert torch.equal(a, b)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("S", [70, 333, 129, 127, 64])      # 127 / 64: the decode steps' own rows are the last / first row of a kv tile
def test_llm_prefill_then_decode_attention(dev, dt, S, hd):
    """llm_qkv_post (half-split RoPE, GQA, append to the row-major K / V caches) + causal prefill attention, then two
    single-token decode steps with the kv length read from device memory."""
    from gar_amd import ops
    from oracle import gar_oracle as O
    B, Hq, Hkv = 2, 4, 2
    Smax = (S + 8 + 63) // 64 * 64
    Wd = (Hq + 2 * Hkv) * hd
    pos = torch.arange(Smax, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = pos[:, None] * inv[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    Kc = torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev)
    Vc = torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev)
    scale = (hd ** -0.5) * 1.4426950408889634

    def ref_qkv(x, p0):
        Sx = x.shape[1]
        xq = x[..., :Hq * hd].view(B, Sx, Hq, hd).transpose(1, 2)
        xk = x[..., Hq * hd:(Hq + Hkv) * hd].view(B, Sx, Hkv, hd).transpose(1, 2)
        xv = x[..., (Hq + Hkv) * hd:].view(B, Sx, Hkv, hd).transpose(1, 2)
        c = torch.cat([cos, cos], -1)[p0:p0 + Sx]
        s_ = torch.cat([sin, sin], -1)[p0:p0 + Sx]
        return xq * c + O._rotate_half(xq) * s_, xk * c + O._rotate_half(xk) * s_, xv

    qkv = q(rnd(B, S, Wd, seed=22), dt)
    Spad = (S + 63) // 64 * 64
    Q = torch.empty(B, Hq, Spad, hd, dtype=dt, device=dev)
    out = torch.empty(B * S, Hq * hd, dtype=dt, device=dev)
    ops.llm_qkv_post(qkv.view(B * S, Wd).to(dev, dt), cos.to(dev), sin.to(dev), Q, Kc, Vc, B, S, Spad, Hq, Hkv, hd, Smax,
                     0, None, scale)
    ops.attention(Q, Kc, Vc, out, B, Hq, Hkv, hd, S, Spad, S, Smax, causal=True, v_row_major=True)
    rq, rk, rv = ref_qkv(qkv, 0)
    rep = Hq // Hkv
    ref = _attn_ref(rq, rk.repeat_interleave(rep, 1), rv.repeat_interleave(rep, 1), True, 0)
    close(out, ref.transpose(1, 2).reshape(B * S, Hq * hd), dt, extra=2.0)
    close(Kc[:, :, :S], rk, dt)
    close(Vc[:, :, :S], rv, dt)
    # decode
    counters = torch.tensor([S, S + 1], dtype=torch.int32, device=dev)
    Q1 = torch.empty(B, Hq, 1, hd, dtype=dt, device=dev)
    o1 = torch.empty(B, Hq * hd, dtype=dt, device=dev)
    ks, vs = [rk], [rv]
    for step in range(2):
        x1 = q(rnd(B, 1, Wd, seed=23 + step), dt)
        K0, V0 = Kc.clone(), Vc.clone()                                           # caches before this step's append
        ops.llm_qkv_post(x1.view(B, Wd).to(dev, dt), cos.to(dev), sin.to(dev), Q1, Kc, Vc, B, 1, 1, Hq, Hkv, hd, Smax,
                         0, counters[0:1], scale)
        ops.attention(Q1, Kc, Vc, o1, B, Hq, Hkv, hd, 1, 1, 0, Smax, causal=False, kv_len_dev=counters[1:2], v_row_major=True)
        o2s = []
        for nsplit in (1, 3, 16):       # split-KV decode kernel (bf16) / routed to the same f32 kernel in parity mode
            ws = torch.empty(ops.attention_decode_workspace(B, Hq, hd, nsplit), dtype=torch.uint8, device=dev)
            o2 = torch.full((B, Hq * hd), float("nan"), dtype=dt, device=dev)
            ops.attention_decode(Q1, Kc, Vc, o2, B, Hq, Hkv, hd, Smax, counters[1:2], nsplit, ws)
            o2s.append(o2)
            # the same step as ONE launch (RoPE + q scale + cache append inside the attention): bit-identical, caches included
            Kf, Vf, o3 = K0.clone(), V0.clone(), torch.full_like(o2, float("nan"))
            took = ops.attention_decode_qkv(x1.view(B, Wd).to(dev, dt), cos.to(dev), sin.to(dev), Kf, Vf, o3, B, Hq, Hkv, hd, Smax,
                                            counters[0:1], scale, nsplit, ws)
            assert took == (dt in HALF)
            if took:
                assert torch.equal(o3, o2) and torch.equal(Kf, Kc) and torch.equal(Vf, Vc)
        ops.counter_add(counters, 1)
        q1, k1, v1 = ref_qkv(x1, S + step)
        ks.append(k1)
        vs.append(v1)
        kk, vv = torch.cat(ks, 2), torch.cat(vs, 2)
        r1 = _attn_ref(q1, kk.repeat_interleave(rep, 1), vv.repeat_interleave(rep, 1), False, 0)
        close(o1, r1.transpose(1, 2).reshape(B, Hq * hd), dt, extra=2.0)
        for o2 in o2s:
            close(o2, r1.transpose(1, 2).reshape(B, Hq * hd), dt, extra=2.0)
    assert counters.tolist() == [S + 2, S + 3]


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd", [64, 128])
def test_llm_left_padded_batch_attention(dev, dt, hd):
    """A LEFT-padded batch (HF generation's layout for prompts of different lengths; the reference forwards
    attention_mask, modeling_gar.py:418-426): `left_pad` shifts the RoPE positions, `kv_start` hides the padding keys in
    the causal prefill and in the decode step (split-KV ranges start at kv_start). Every sequence's real rows must equal
    the same sequence run alone and unpadded; pads of 0, inside a tile, exactly one tile, and more than a 128-row q block."""
    from gar_amd import ops
    from oracle import gar_oracle as O
    S, Hq, Hkv = 333, 4, 2
    pads = [0, 37, 64, 150, 301]
    B = len(pads)
    Smax = (S + 8 + 63) // 64 * 64
    Wd = (Hq + 2 * Hkv) * hd
    pos = torch.arange(Smax, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = pos[:, None] * inv[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    scale = (hd ** -0.5) * 1.4426950408889634
    rep = Hq // Hkv

    def ref_qkv(x, p0):                       # x [1, Sx, Wd] of ONE sequence at positions p0 ..
        Sx = x.shape[1]
        xq = x[..., :Hq * hd].view(1, Sx, Hq, hd).transpose(1, 2)
        xk = x[..., Hq * hd:(Hq + Hkv) * hd].view(1, Sx, Hkv, hd).transpose(1, 2)
        xv = x[..., (Hq + Hkv) * hd:].view(1, Sx, Hkv, hd).transpose(1, 2)
        c = torch.cat([cos, cos], -1)[p0:p0 + Sx]
        s_ = torch.cat([sin, sin], -1)[p0:p0 + Sx]
        return xq * c + O._rotate_half(xq) * s_, xk * c + O._rotate_half(xk) * s_, xv

    qkv = q(rnd(B, S, Wd, seed=41), dt)
    Spad = (S + 63) // 64 * 64
    Kc = torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev)
    Vc = torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev)
    Q = torch.empty(B, Hq, Spad, hd, dtype=dt, device=dev)
    out = torch.full((B * S, Hq * hd), float("nan"), dtype=dt, device=dev)
    lp = torch.tensor(pads, dtype=torch.int32, device=dev)
    ops.llm_qkv_post(qkv.view(B * S, Wd).to(dev, dt), cos.to(dev), sin.to(dev), Q, Kc, Vc, B, S, Spad, Hq, Hkv, hd, Smax,
                     0, None, scale, left_pad=lp)
    ops.attention(Q, Kc, Vc, out, B, Hq, Hkv, hd, S, Spad, S, Smax, causal=True, kv_start=lp, v_row_major=True)
    assert torch.isfinite(out.float()).all()               # padding rows too: they flow through the following GEMMs
    outv = out.view(B, S, Hq * hd)
    refs = []
    for b, pd in enumerate(pads):
        rq, rk, rv = ref_qkv(qkv[b:b + 1, pd:], 0)
        ref = _attn_ref(rq, rk.repeat_interleave(rep, 1), rv.repeat_interleave(rep, 1), True, 0)
        close(outv[b, pd:], ref.transpose(1, 2).reshape(S - pd, Hq * hd), dt, extra=2.0)
        close(Kc[b, :, pd:S], rk[0], dt)
        close(Vc[b, :, pd:S], rv[0], dt)
        refs.append(([rk], [rv]))
    counters = torch.tensor([S, S + 1], dtype=torch.int32, device=dev)
    Q1 = torch.empty(B, Hq, 1, hd, dtype=dt, device=dev)
    for step in range(2):
        x1 = q(rnd(B, 1, Wd, seed=43 + step), dt)
        K0, V0 = Kc.clone(), Vc.clone()
        ops.llm_qkv_post(x1.view(B, Wd).to(dev, dt), cos.to(dev), sin.to(dev), Q1, Kc, Vc, B, 1, 1, Hq, Hkv, hd, Smax,
                         0, counters[0:1], scale, left_pad=lp)
        outs = []
        for nsplit in (1, 3, 16):
            ws = torch.empty(ops.attention_decode_workspace(B, Hq, hd, nsplit), dtype=torch.uint8, device=dev)
            o2 = torch.full((B, Hq * hd), float("nan"), dtype=dt, device=dev)
            ops.attention_decode(Q1, Kc, Vc, o2, B, Hq, Hkv, hd, Smax, counters[1:2], nsplit, ws, kv_start=lp)
            outs.append(o2)
            Kf, Vf, o3 = K0.clone(), V0.clone(), torch.full_like(o2, float("nan"))
            if ops.attention_decode_qkv(x1.view(B, Wd).to(dev, dt), cos.to(dev), sin.to(dev), Kf, Vf, o3, B, Hq, Hkv, hd, Smax,
                                        counters[0:1], scale, nsplit, ws, left_pad=lp):
                assert torch.equal(o3, o2) and torch.equal(Kf, Kc) and torch.equal(Vf, Vc)
            else:
                assert dt == torch.float32
        ops.counter_add(counters, 1)
        for b, pd in enumerate(pads):
            q1, k1, v1 = ref_qkv(x1[b:b + 1], S - pd + step)
            refs[b][0].append(k1)
            refs[b][1].append(v1)
            kk, vv = torch.cat(refs[b][0], 2), torch.cat(refs[b][1], 2)
            r1 = _attn_ref(q1, kk.repeat_interleave(rep, 1), vv.repeat_interleave(rep, 1), False, 0)
            for o2 in outs:
                close(o2[b:b + 1], r1.transpose(1, 2).reshape(1, Hq * hd), dt, extra=2.0)


# ------------------------------------------------------------------------------------------------------------------
def test_placeholder_scan_and_assemble(dev):
    from gar_amd import ops
    B, S, Cc, V = 2, 2500, 64, 400
    g = torch.Generator().manual_seed(30)
    ids = torch.randint(0, 290, (B, S), generator=g)
    img_tok, crops = 300, [304, 305, 308, 310, 311]
    ids[0, 5:5 + 1200] = img_tok
    ids[1, 100:900] = img_tok
    ids[1, 1000:1400] = img_tok
    ids[0, 1300:1316] = 305
    ids[1, 1500:1516] = 310
    ids[1, 1600:1616] = 304
    slot = torch.empty(B, S, dtype=torch.int32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    spans = torch.empty(B, 5, 2, dtype=torch.int32, device=dev)
    ops.placeholder_scan(ids.to(dev), img_tok, torch.tensor(crops, device=dev), slot, counts, spans)
    m = ids == img_tok
    ref_slot = torch.where(m, m.cumsum(1) - 1, torch.full_like(ids, -1)).int()
    assert torch.equal(slot.cpu(), ref_slot)
    assert counts.tolist() == [1200, 1200]
    assert spans.cpu().tolist() == [[[-1, -1], [1300, 1315], [-1, -1], [-1, -1], [-1, -1]],
                                    [[1600, 1615], [-1, -1], [-1, -1], [1500, 1515], [-1, -1]]]
    for dt in DT:
        E, feats = q(rnd(V, Cc, seed=31), dt), q(rnd(B, 1200, Cc, seed=32), dt)
        out = torch.empty(B, S, Cc, dtype=dt, device=dev)
        ops.embed_assemble(ids.to(dev), slot, E.to(dev, dt), feats.to(dev, dt), out, 1200)
        ref = F.embedding(ids, E)
        ref = ref.masked_scatter(m.unsqueeze(-1).expand_as(ref), feats)
        assert torch.equal(out.float().cpu(), ref)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("bbox", [(0.720703125, 0.8688311688311688, 0.7939453125, 0.9233766233766234),
                                  (0.0, 0.0, 1.0, 1.0), (0.3, 0.55, 0.31, 0.56), (0.02, 0.5, 0.35, 0.78)])
def test_roi_replay_bit_exact_vs_oracle(dev, dt, bbox):
    """replay kernel == oracle feature_replay (merge + fp32 roi_align + cast + splice), bit for bit."""
    from gar_amd import GARConfig, ops
    from oracle import gar_oracle as O
    cfg = GARConfig.gar_1b()
    P, Cc, ncw, nch = 16, 64, 3, 2
    tiles = ncw * nch + 1
    feats = q(rnd(tiles, P * P, Cc, seed=33), dt)
    S = 700
    ids = torch.full((1, S), 7, dtype=torch.int64)
    ids[0, 100:356] = 128005
    emb = q(rnd(1, S, Cc, seed=34), dt)
    ref = O.feature_replay(emb.to(dt), ids, feats.to(dt), torch.tensor([[ncw, nch]]), [{"128005": bbox}], cfg)
    spans = torch.tensor([[-1, -1], [100, 355], [-1, -1], [-1, -1], [-1, -1]], dtype=torch.int32, device=dev)
    e = emb.to(dev, dt).clone()
    roi, ss = O.replay_roi(bbox, P * nch, P * ncw, cfg.feat_stride)
    ops.roi_replay(feats.to(dev, dt), e[0], spans, 1, 1, ncw, nch, P, Cc, S, roi[1:], ss, 2, True)
    assert torch.equal(e.float().cpu(), ref.float())
    # absent crop token: nothing is written
    e2 = emb.to(dev, dt).clone()
    ops.roi_replay(feats.to(dev, dt), e2[0], spans, 0, 1, ncw, nch, P, Cc, S, roi[1:], ss, 2, True)
    assert torch.equal(e2.float().cpu(), emb)


def _kat_cases():
    import parity_util as PU
    return PU.roi_kat_cases()


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("case", _kat_cases(), ids=lambda c: c[0])
def test_roi_replay_hand_derived_known_answers(dev, dt, case):
    """The replay kernel against the hand-derived roi_align known answers of tests/parity_util.py (aligned half-pixel
    shift, the two sampling_ratio=2 sample positions at a non-integer bin size, the y <= 0 clamp, the last-cell branch,
    out-of-range samples, a box straddling two tiles of the tile layout, the reference's double scaling) — the same
    closed forms the numpy oracle and its C twin are held to on the CPU. The thumbnail tile is filled with 1e9 and must
    never be read. The map cells the samples touch are exactly representable in bf16; bf16 outputs round once."""
    import parity_util as PU
    from gar_amd import ops
    name, ncw, nch, chans, roi, ss, _ = case
    fmap, exp = PU.kat_feature_map(case)
    C, Cc, P, S = fmap.shape[0], 64, 16, 300
    tiles = torch.zeros(1 + ncw * nch, P * P, Cc)
    tiles[:, :, :C] = PU.tiles_from_map(fmap, ncw, nch)
    tiles[0] = 1.0e9
    emb = torch.full((S, Cc), -7.0)
    spans = torch.tensor([[-1, -1], [20, 275], [-1, -1], [-1, -1], [-1, -1]], dtype=torch.int32, device=dev)
    e = emb.to(dev, dt).clone()
    ops.roi_replay(tiles.to(dev, dt), e, spans, 1, 1, ncw, nch, P, Cc, S, roi, ss, 2, True)
    out = e.float().cpu()
    assert float((out[:20] + 7.0).abs().max()) == 0 and float((out[276:] + 7.0).abs().max()) == 0
    got = out[20:276, :C].double().T.reshape(C, P, P)
    lim = 2e-5 if dt == torch.float32 else 4e-3 * max(1.0, float(exp.abs().max()))
    assert float((got - exp).abs().max()) < lim, (name, float((got - exp).abs().max()))
    assert float(out[20:276, C:].abs().max()) == 0


@pytest.mark.parametrize("dt", DT)
def test_roi_replay_batched_equals_per_token_launches(dev, dt):
    """one launch over (sample, crop token) jobs == the per-token kernel, bit for bit; ragged: sample 1 has two crop
    tokens with different canvases per sample, sample 2 none, one job points at an absent token."""
    from gar_amd import GARConfig, ops
    from oracle import gar_oracle as O
    cfg = GARConfig.gar_1b()
    P, Cc, B, tiles, S, ncrop = 16, 128, 3, 7, 900, 5
    feats = q(rnd(B * tiles, P * P, Cc, seed=36), dt).to(dev, dt)
    emb = q(rnd(B, S, Cc, seed=37), dt).to(dev, dt)
    spans = torch.full((B, ncrop, 2), -1, dtype=torch.int32)
    spans[0, 1] = torch.tensor([10, 265])
    spans[1, 0] = torch.tensor([300, 555])
    spans[1, 3] = torch.tensor([600, 855])
    spans = spans.to(dev)
    canv = {0: (3, 2), 1: (2, 3)}
    boxes = {(0, 1): (0.1, 0.2, 0.6, 0.9), (1, 0): (0.0, 0.0, 1.0, 1.0), (1, 3): (0.72, 0.87, 0.79, 0.92),
             (1, 4): (0.2, 0.2, 0.4, 0.4)}                     # (1, 4): bbox present but token absent from input_ids
    jobs, ref = [], emb.clone()
    for (b, ci), bbox in boxes.items():
        ncw, nch = canv[b]
        roi, ss = O.replay_roi(bbox, P * nch, P * ncw, cfg.feat_stride)
        jobs.append((b, ci, 1, ncw, nch, *roi[1:], ss))
        ops.roi_replay(feats[b * tiles:(b + 1) * tiles], ref[b], spans[b], ci, 1, ncw, nch, P, Cc, S, roi[1:], ss, 2, True)
    out = emb.clone()
    ops.roi_replay_batched(feats, out, spans, ops.roi_jobs_tensor(jobs, dev), ncrop, tiles, P, Cc, S, 2, True)
    assert torch.equal(out.cpu(), ref.cpu())
    assert not torch.equal(out.cpu(), emb.cpu())
    assert torch.equal(out[2].cpu(), emb[2].cpu())


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("npt", [0, 1])
def test_pool_assemble_inplace_replay_equal_the_unfused_pass(dev, dt, npt):
    """pool_assemble + roi_replay_inplace (what generate() runs: the pooled features live only inside the sequence) ==
    pool2x2 -> embed_assemble -> roi_replay_batched, bit for bit — with the image placeholders split into two runs
    (rank_pos must not assume a contiguous block), a cls row to skip (npt = 1), two samples with different canvases."""
    from gar_amd import GARConfig, ops
    from oracle import gar_oracle as O
    cfg = GARConfig.gar_1b()
    P, g, Cc, B, tiles, V, ncrop = 16, 32, 128, 2, 7, 500, 5
    N = g * g + npt
    n_rows = tiles * P * P
    S = n_rows + 256 + 40
    proj = q(rnd(B * tiles * N, Cc, seed=40), dt).to(dev, dt)
    E = q(rnd(V, Cc, seed=41), dt).to(dev, dt)
    ids = torch.randint(0, 290, (B, S), generator=torch.Generator().manual_seed(42))
    img_tok, crops = 300, [304, 305, 308, 310, 311]
    ids[0, 3:3 + 1000] = img_tok                       # sample 0: two runs of placeholders
    ids[0, 1010:1010 + n_rows - 1000] = img_tok
    ids[0, S - 270:S - 14] = 305
    ids[1, 10:10 + n_rows] = img_tok                   # sample 1: one run
    ids[1, S - 280:S - 24] = 308
    assert int((ids[0] == img_tok).sum()) == n_rows and int((ids[1] == img_tok).sum()) == n_rows
    ids = ids.to(dev)
    slot = torch.empty(B, S, dtype=torch.int32, device=dev)
    counts = torch.empty(B, dtype=torch.int32, device=dev)
    spans = torch.empty(B, ncrop, 2, dtype=torch.int32, device=dev)
    rank_pos = torch.zeros(B, n_rows, dtype=torch.int32, device=dev)
    ops.placeholder_scan(ids, img_tok, torch.tensor(crops, device=dev), slot, counts, spans, rank_pos)
    pos = (ids == img_tok).nonzero()
    assert torch.equal(rank_pos[0].cpu().long(), pos[pos[:, 0] == 0, 1].cpu())
    canv = {0: (3, 2), 1: (2, 3)}
    boxes = {(0, 1): (0.1, 0.2, 0.6, 0.9), (1, 2): (0.72, 0.87, 0.79, 0.92)}
    jobs = []
    for (b, ci), bbox in boxes.items():
        ncw, nch = canv[b]
        roi, ss = O.replay_roi(bbox, P * nch, P * ncw, cfg.feat_stride)
        jobs.append((b, ci, 1, ncw, nch, *roi[1:], ss))
    jt = ops.roi_jobs_tensor(jobs, dev)
    # unfused
    feats = torch.empty(B * tiles, P * P, Cc, dtype=dt, device=dev)
    ops.pool2x2(proj, feats, g, in_tile_tokens=N, in_token_offset=npt)
    ref = torch.empty(B, S, Cc, dtype=dt, device=dev)
    ops.embed_assemble(ids, slot, E, feats, ref, n_rows)
    plain = ref.clone()
    ops.roi_replay_batched(feats, ref, spans, jt, ncrop, tiles, P, Cc, S, 2, True)
    # fused
    out = torch.full((B, S, Cc), 7.0, dtype=dt, device=dev)
    ops.pool_assemble(ids, slot, E, proj, out, tiles, g, N, npt)
    assert torch.equal(out.cpu(), plain.cpu())
    ops.roi_replay_inplace(out, spans, rank_pos, jt, ncrop, P, Cc, S, 2, True)
    assert torch.equal(out.cpu(), ref.cpu()) and not torch.equal(ref.cpu(), plain.cpu())


@pytest.mark.parametrize("dt", DT)
def test_argmax_first_index_tiebreak_and_lookup(dev, dt):
    from gar_amd import ops
    B, V, Cc = 3, 128262, 128
    lg = q(rnd(B, V, seed=35), dt)
    lg[0, 777] = 50.0
    lg[0, 90000] = 50.0           # tie -> first index
    lg[1, V - 1] = 60.0
    lg[2, 0] = 70.0
    ld = (V + 63) // 64 * 64
    L = torch.full((B, ld), 1000.0, dtype=dt, device=dev)   # padding columns must be ignored
    L[:, :V] = lg.to(dev, dt)
    out = torch.zeros(B, 4, dtype=torch.int64, device=dev)
    cur = torch.zeros(B, dtype=torch.int64, device=dev)
    step = torch.tensor([2], dtype=torch.int32, device=dev)
    ws = torch.empty(ops.argmax_workspace(B, V), dtype=torch.uint8, device=dev)
    ops.argmax(L, V, out, 4, step, cur, ws)
    assert cur.tolist() == [777, V - 1, 0] and out[:, 2].tolist() == [777, V - 1, 0]
    assert torch.equal(cur.cpu(), lg.argmax(-1))
    E = q(rnd(1000, Cc, seed=36), dt)
    h = torch.empty(B, Cc, dtype=dt, device=dev)
    ops.embed_lookup(torch.tensor([5, 999, 0], device=dev), E.to(dev, dt), h)
    assert torch.equal(h.float().cpu(), E[[5, 999, 0]])


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("epi", ["none", "swiglu"])
def test_skinny_gemm_with_fused_rmsnorm(dev, dt, epi):
    """decode path: RMSNorm folded into the weight-streaming GEMM (x*g as operand, rstd applied to the accumulator)."""
    from gar_amd import hip, ops
    M, K, N = 6, 2048, 512
    x = q(rnd(M, K, seed=40, scale=3.0), dt)
    g = q(1 + 0.1 * rnd(K, seed=41), dt)
    w = q(rnd(N, K, seed=42, scale=K ** -0.5), dt)
    xd = x.double()
    n = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5) * g.double()
    if epi == "none":
        out = torch.empty(M, N, dtype=dt, device=dev)
        ops.gemm(x.to(dev, dt), w.to(dev, dt), out, norm_w=g.to(dev, dt), norm_eps=1e-5)
        close(out, n @ w.double().T, dt, extra=2.0)
    else:
        Fd = N // 2
        gw, uw = w[:Fd], w[Fd:]
        gu = torch.stack([gw.view(Fd // 16, 16, K), uw.view(Fd // 16, 16, K)], 1).reshape(N, K)
        out = torch.empty(M, Fd, dtype=dt, device=dev)
        ops.gemm(x.to(dev, dt), gu.to(dev, dt), out, hip.EPI_SWIGLU, norm_w=g.to(dev, dt), norm_eps=1e-5)
        close(out, F.silu(n @ gw.double().T) * (n @ uw.double().T), dt, extra=2.0)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("M", [3, 17, 33, 64])
@pytest.mark.parametrize("N", [768, 8192 + 64])
def test_skinny_gemm_with_folded_rmsnorm(dev, M, N, dt):
    """decode path, gar_gemm_params.norm_folded: W carries the RMSNorm gain, the kernel takes the row sums of squares of the
    activations off the matrix pipe (diagonal of x_tile x_tile^T) and scales the accumulator rows: equals the fp64
    rmsnorm -> linear (-> SwiGLU) of the same bf16 activations; narrow (1-2 weight tiles per block) and wide (4) outputs."""
    from gar_amd import hip, ops
    K = 2048
    x = q(rnd(M, K, seed=140, scale=2.5) + 0.3, dt)
    g = 1 + 0.1 * rnd(K, seed=141)
    w = rnd(N, K, seed=142, scale=K ** -0.5)
    wf = q(w * g[None, :], dt)                                     # W diag(g), rounded once
    xd = x.double()
    y = (xd @ wf.double().T) * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5)
    out = torch.empty(M, N, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), wf.to(dev, dt), out, norm_folded=True, norm_eps=1e-5)
    close(out, y, dt, extra=1.5)
    Fd = N // 2
    gu = torch.stack([wf[:Fd].view(Fd // 16, 16, K), wf[Fd:].view(Fd // 16, 16, K)], 1).reshape(N, K)
    o2 = torch.empty(M, Fd, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), gu.to(dev, dt), o2, hip.EPI_SWIGLU, norm_folded=True, norm_eps=1e-5)
    close(o2, F.silu(y[:, :Fd]) * y[:, Fd:], dt, extra=2.0)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("M", [1, 7, 16])
@pytest.mark.parametrize("N", [8, 24, 2048])
@pytest.mark.parametrize("form", ["plain", "norm_w", "norm_folded", "bias", "res"])
def test_skinny_gemm_half_tiles(dev, M, N, form, dt):
    """M <= 16 rows and N <= 2048 columns take the EIGHT-row half-tile blocks of the skinny GEMM (ROWS = 8: the upper half of a
    block's MFMA tile is zero, its output rows are never stored — ADVICE r4): every prologue / epilogue the decode step uses on
    that path against the fp64 reference, with the LDS poisoned first (a launch that fills the CU's LDS with NaN bit patterns:
    rows 8..15 of a weight slot must not leak into the rows that are stored)."""
    from gar_amd import hip, ops
    K = 512
    x = q(rnd(M, K, seed=240, scale=2.0) + 0.2, dt)
    w = q(rnd(N, K, seed=241, scale=K ** -0.5), dt)
    g = q(1 + 0.1 * rnd(K, seed=242), dt)
    bias = q(rnd(N, seed=243), dt)
    res = q(rnd(M, N, seed=244), dt)
    xd, wd = x.double(), w.double()
    rstd = torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5)
    # poison: an attention launch that DMA-stages NaN rows through the LDS of every CU
    nan = torch.full((1, 64, 256, 64), float("nan"), dtype=dt, device=dev)
    junk = torch.empty(256, 64 * 64, dtype=dt, device=dev)
    ops.attention(nan, nan, nan, junk, 1, 64, 64, 64, 256, 256, 256, 256, causal=False, v_row_major=True)
    out = torch.full((M, N), 7.0, dtype=dt, device=dev)
    xg, wg = x.to(dev, dt), w.to(dev, dt)
    if form == "plain":
        ops.gemm(xg, wg, out)
        ref = xd @ wd.T
    elif form == "norm_w":
        ops.gemm(xg, wg, out, norm_w=g.to(dev, dt), norm_eps=1e-5)
        ref = (xd * rstd * g.double()) @ wd.T
    elif form == "norm_folded":
        ops.gemm(xg, wg, out, norm_folded=True, norm_eps=1e-5)
        ref = (xd @ wd.T) * rstd
    elif form == "bias":
        ops.gemm(xg, wg, out, hip.EPI_BIAS, bias=bias.to(dev, dt))
        ref = xd @ wd.T + bias.double()
    else:
        out.copy_(res.to(dev, dt))
        ops.gemm(xg, wg, out, hip.EPI_RES, residual=out)
        ref = xd @ wd.T + res.double()
    assert torch.isfinite(out.float()).all()
    close(out, ref, dt, extra=2.0)


@pytest.mark.parametrize("dt", HALF)
def test_gemm_norm_folded_refuses_large_m(dev, dt):
    from gar_amd import hip, ops
    with pytest.raises(hip.GarError, match="norm_folded"):
        ops.gemm(torch.zeros(65, 128, dtype=dt, device=dev), torch.zeros(64, 128, dtype=dt, device=dev),
                 torch.zeros(65, 64, dtype=dt, device=dev), norm_folded=True, norm_eps=1e-5)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("M", [17, 32, 40, 64])
def test_skinny_gemm_batched_rows_bf16(dev, M, dt):
    """decode GEMMs for up to 64 batched sequences: 2 / 4 accumulator row tiles per weight tile, all decode epilogues,
    with and without the fused RMSNorm."""
    from gar_amd import hip, ops
    K, N = 2048, 768
    x = q(rnd(M, K, seed=60, scale=2.0), dt)
    g = q(1 + 0.1 * rnd(K, seed=61), dt)
    w = q(rnd(N, K, seed=62, scale=K ** -0.5), dt)
    res = q(rnd(M, N, seed=63), dt)
    xd, wd = x.double(), w.double()
    nrm = xd * torch.rsqrt(xd.pow(2).mean(-1, keepdim=True) + 1e-5) * g.double()
    out = torch.empty(M, N, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), w.to(dev, dt), out)
    close(out, xd @ wd.T, dt)
    ops.gemm(x.to(dev, dt), w.to(dev, dt), out, norm_w=g.to(dev, dt), norm_eps=1e-5)
    close(out, nrm @ wd.T, dt, extra=2.0)
    r = res.to(dev, dt).clone()
    ops.gemm(x.to(dev, dt), w.to(dev, dt), r, hip.EPI_RES, residual=r)
    close(r, res.double() + xd @ wd.T, dt)
    Fd = N // 2
    gw, uw = w[:Fd], w[Fd:]
    gu = torch.stack([gw.view(Fd // 16, 16, K), uw.view(Fd // 16, 16, K)], 1).reshape(N, K)
    o2 = torch.empty(M, Fd, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), gu.to(dev, dt), o2, hip.EPI_SWIGLU, norm_w=g.to(dev, dt), norm_eps=1e-5)
    close(o2, F.silu(nrm @ gw.double().T) * (nrm @ uw.double().T), dt, extra=2.0)
    # wide outputs (gate/up, lm_head) take 4 weight tiles per block; N with a ragged last block
    Nw = 8192 + 48
    ww = q(rnd(Nw, K, seed=64, scale=K ** -0.5), dt)
    ow = torch.empty(M, Nw, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), ww.to(dev, dt), ow)
    close(ow, xd @ ww.double().T, dt)
    Fw = 4096 + 16
    gw2, uw2 = ww[:Fw], ww[Fw:2 * Fw]
    gu2 = torch.stack([gw2.view(Fw // 16, 16, K), uw2.view(Fw // 16, 16, K)], 1).reshape(2 * Fw, K)
    o3 = torch.empty(M, Fw, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), gu2.to(dev, dt), o3, hip.EPI_SWIGLU)
    close(o3, F.silu(xd @ gw2.double().T) * (xd @ uw2.double().T), dt)
    # more 16-row tiles than CUs but not "wide" (Llama-3.1-8B qkv: N = 6144): two weight tiles per block for M > 32, with
    # a ragged last block; plain, residual and the weight ring of the staged path (4 K steps in flight per wave)
    Nm = 6144 + 16
    wm = ww[:Nm]
    om = torch.empty(M, Nm, dtype=dt, device=dev)
    ops.gemm(x.to(dev, dt), wm.to(dev, dt), om)
    close(om, xd @ wm.double().T, dt)
    resm = q(rnd(M, Nm, seed=65), dt)
    rm = resm.to(dev, dt).clone()
    ops.gemm(x.to(dev, dt), wm.to(dev, dt), rm, hip.EPI_RES, residual=rm)
    close(rm, resm.double() + xd @ wm.double().T, dt)
    # a row's result does not depend on the batch it rides in (up to the split-K summation order)
    o1 = torch.empty(1, N, dtype=dt, device=dev)
    ops.gemm(x[M - 1:M].to(dev, dt).contiguous(), w.to(dev, dt), o1)
    ops.gemm(x.to(dev, dt), w.to(dev, dt), out)
    close(o1[0], out[M - 1].float(), dt)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("M", [17, 40, 64])
@pytest.mark.parametrize("S", [2, 4, 8])
def test_decode_gemm_split_k_and_fused_reduce(dev, M, S, dt):
    """Llama `down` at decode time: K slices of the GEMM written as fp32 partials (gar_gemm split_k) and reduced, together
    with the residual add and the RMSNorm that follows, by gar_splitk_residual_rmsnorm — against fp64, and against the
    unfused pair gemm(EPI_RES) + rmsnorm it replaces (same roundings; only the fp32 summation order of the slices
    differs)."""
    from gar_amd import hip, ops
    K, N = 8192, 2048
    x = q(rnd(M, K, seed=80, scale=1.5), dt)
    w = q(rnd(N, K, seed=81, scale=K ** -0.5), dt)
    h0 = q(rnd(M, N, seed=82), dt)
    g = q(1 + 0.1 * rnd(N, seed=83), dt)
    xg, wg, gg = x.to(dev, dt), w.to(dev, dt), g.to(dev, dt)
    partial = torch.full((S, M, N), float("nan"), dtype=torch.float32, device=dev)
    ops.gemm(xg, wg, None, partial=partial)
    prod = x.double() @ w.double().T
    close(partial.sum(0), prod, torch.float32, extra=50.0)            # fp32 accumulation of bf16 products over K = 8192
    for s_ in range(S):                                               # every slice is its own K range
        ks = K // S
        close(partial[s_], x[:, s_ * ks:(s_ + 1) * ks].double() @ w[:, s_ * ks:(s_ + 1) * ks].double().T, torch.float32,
              extra=50.0)
    h = h0.to(dev, dt).clone()
    y = torch.empty(M, N, dtype=dt, device=dev)
    ops.splitk_residual_rmsnorm(partial, h, gg, 1e-5, out=y)
    href = h0.double() + prod
    close(h, href, dt)
    hr = h.float().cpu().double()                                     # the norm reads the rounded residual stream
    close(y, hr * torch.rsqrt(hr.pow(2).mean(-1, keepdim=True) + 1e-5) * g.double(), dt, extra=2.0)
    # the unfused pair
    h1 = h0.to(dev, dt).clone()
    ops.gemm(xg, wg, h1, hip.EPI_RES, residual=h1)
    y1 = torch.empty(M, N, dtype=dt, device=dev)
    ops.rmsnorm(h1, gg, 1e-5, out=y1)
    ulp = (h.float() - h1.float()).abs() / h1.float().abs().clamp_min(1e-3)
    assert float(ulp.max()) <= 2 ** -7 and float((h != h1).float().mean()) < 0.02      # rare 1-ulp flips only
    same = (h == h1).all(-1)
    # identical rows in -> identical rows out (fp16's ulp is 8x finer: a row of 2048 without a single flip may not exist)
    assert (bool(same.any()) or dt == torch.float16) and torch.equal(y[same], y1[same])
    # residual update only (no norm output)
    h2 = h0.to(dev, dt).clone()
    ops.splitk_residual_rmsnorm(partial, h2, None, 1e-5, out=None)
    assert torch.equal(h2, h)
    with pytest.raises(hip.GarError):
        ops.gemm(xg, wg, None, hip.EPI_RES, residual=h1, partial=partial)


def _fold_ln(W, b, gamma, beta):
    """LN(x) W^T + b = rstd * (x Wc^T) + b':  Wc = W diag(gamma) minus each row's mean,  b' = b + W beta"""
    Wg = W.double() * gamma.double()[None, :]
    return (Wg - Wg.mean(1, keepdim=True)).float(), (b.double() + W.double() @ beta.double()).float()


def test_row_rstd_and_stats_finalize(dev):
    from gar_amd import ops
    for M, D in ((1000, 1024), (77, 2560), (3, 64)):          # 16 / 40 / 1 strips of 64 columns; rows not a multiple of 16
        x = q(rnd(M, D, seed=90) * 2.0 + 0.7, torch.bfloat16)
        xd = x.to(dev, torch.bfloat16)
        for rms in (False, True):
            out = torch.empty(M, dtype=torch.float32, device=dev)
            ops.row_rstd(xd, 1e-5, rms, out)
            var = x.double().pow(2).mean(1) if rms else x.double().var(1, unbiased=False)
            ref = (var + 1e-5).rsqrt()
            assert float((out.cpu().double() - ref).abs().max() / ref.abs().max()) < 2e-6
            # the same from per-strip partials
            st = torch.stack([x.double().view(M, D // 64, 64).sum(-1), x.double().view(M, D // 64, 64).pow(2).sum(-1)], -1).float()
            out2 = torch.full((M,), float("nan"), dtype=torch.float32, device=dev)
            ops.row_stats_finalize(st.to(dev).contiguous(), D, 1e-5, rms, out2)
            assert float((out2.cpu().double() - ref).abs().max() / ref.abs().max()) < 1e-4


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("epi", ["bias", "gelu", "none_rms", "swiglu_rms"])
def test_gemm_row_scale_is_the_folded_norm(dev, epi, dt):
    """gar_gemm_params.row_scale: the norm in front of a GEMM folded into it — x goes in un-normalised, the weight is
    W diag(gamma) with centred rows (LayerNorm) or W diag(g) (RMSNorm), the accumulator rows are scaled by rstd[m] and the
    folded bias is added after the scale. Against the fp64 norm -> linear (-> GELU / SwiGLU) of the same bf16 inputs."""
    from gar_amd import hip, ops
    M, K, N = 8300, 1024, 1024
    x = q(rnd(M, K, seed=91) * 1.5 + 0.4, dt)
    W = rnd(N, K, seed=92, scale=K ** -0.5)
    gamma, beta, b = 1.0 + 0.1 * rnd(K, seed=93), 0.1 * rnd(K, seed=94), 0.1 * rnd(N, seed=95)
    xd = x.to(dev, dt)
    rstd = torch.empty(M, dtype=torch.float32, device=dev)
    rms = epi.endswith("rms")
    ops.row_rstd(xd, 1e-5, rms, rstd)
    xs = x.double()
    if rms:
        Wf = q(W * gamma[None, :], dt)
        normed = xs * (xs.pow(2).mean(1, keepdim=True) + 1e-5).rsqrt()
        ref = normed @ Wf.double().T                    # gamma folded: compare against the folded (rounded) weight itself
        bf = None
    else:
        Wc, bp = _fold_ln(W, b, gamma, beta)
        Wf, bf = q(Wc, dt), q(bp, dt)
        normed = (xs - xs.mean(1, keepdim=True)) * (xs.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt()
        # what the fold computes exactly: rstd * (x Wf^T) + b'; and that it IS the LayerNorm -> linear (up to Wf's rounding)
        ref = (xs @ Wf.double().T) * (xs.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() + bf.double()
        ln_lin = (normed * gamma.double() + beta.double()) @ W.double().T + b.double()
        assert float((ref - ln_lin).abs().max() / ln_lin.abs().max()) < 1e-2
    if epi == "bias":
        out = torch.empty(M, N, dtype=dt, device=dev)
        ops.gemm(xd, Wf.to(dev, dt), out, hip.EPI_BIAS, bias=bf.to(dev, dt), row_scale=rstd)
    elif epi == "gelu":
        out = torch.empty(M, N, dtype=dt, device=dev)
        ops.gemm(xd, Wf.to(dev, dt), out, hip.EPI_BIAS_GELU, bias=bf.to(dev, dt), row_scale=rstd)
        ref = torch.nn.functional.gelu(ref)
    elif epi == "none_rms":
        out = torch.empty(M, N, dtype=dt, device=dev)
        ops.gemm(xd, Wf.to(dev, dt), out, hip.EPI_NONE, row_scale=rstd)
    else:
        out = torch.empty(M, N // 2, dtype=dt, device=dev)
        ops.gemm(xd, Wf.to(dev, dt), out, hip.EPI_SWIGLU, row_scale=rstd)
        r = ref.view(M, N // 32, 2, 16)                  # [gate16 | up16] row blocks
        ref = (torch.nn.functional.silu(r[:, :, 0]) * r[:, :, 1]).reshape(M, N // 2)
    close(out, ref, dt, extra=1.5)
    # small problems do not take the tile GEMM: refused, not silently unscaled
    with pytest.raises(hip.GarError):
        ops.gemm(xd[:64], Wf.to(dev, dt), torch.empty(64, N, dtype=dt, device=dev), hip.EPI_NONE, row_scale=rstd)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("epi", ["res", "bsr"])
def test_gemm_row_stats_of_the_rounded_outputs(dev, epi, dt):
    """gar_gemm_params.row_stats: the producer of a folded norm writes (sum, sum of squares) of its bf16-ROUNDED output rows
    per 64-column strip; finalize gives the rstd a LayerNorm / RMSNorm of that output would compute. M and N tails."""
    from gar_amd import hip, ops
    M, K, N = 8300, 256, 1000
    a = q(rnd(M, K, seed=96), dt).to(dev, dt)
    w = q(rnd(N, K, seed=97, scale=K ** -0.5), dt).to(dev, dt)
    res = q(rnd(M, N, seed=98), dt).to(dev, dt)
    out = torch.empty(M, N, dtype=dt, device=dev)
    strips = (N + 63) // 64
    st = torch.full((M, strips, 2), float("nan"), dtype=torch.float32, device=dev)
    if epi == "res":
        ops.gemm(a, w, out, hip.EPI_RES, residual=res, row_stats=st)
        plain = torch.empty_like(out)
        ops.gemm(a, w, plain, hip.EPI_RES, residual=res)
    else:
        bias, gam = q(rnd(N, seed=99), dt).to(dev, dt), q(0.1 + 0.01 * rnd(N, seed=100), dt).to(dev, dt)
        ops.gemm(a, w, out, hip.EPI_BIAS_SCALE_RES, bias=bias, residual=res, gamma=gam, row_stats=st)
        plain = torch.empty_like(out)
        ops.gemm(a, w, plain, hip.EPI_BIAS_SCALE_RES, bias=bias, residual=res, gamma=gam)
    assert torch.equal(out, plain)                                   # the statistics do not change the output
    o = out.double().cpu()
    pad = torch.zeros(M, strips * 64, dtype=torch.float64)
    pad[:, :N] = o
    ref = torch.stack([pad.view(M, strips, 64).sum(-1), pad.view(M, strips, 64).pow(2).sum(-1)], -1)
    got = st.cpu().double()
    assert torch.isfinite(got).all()
    assert float((got - ref).abs().max() / ref.abs().max()) < 1e-5
    for rms in (False, True):
        r = torch.empty(M, dtype=torch.float32, device=dev)
        ops.row_stats_finalize(st, N, 1e-5, rms, r)
        var = o.pow(2).mean(1) if rms else o.var(1, unbiased=False)
        want = (var + 1e-5).rsqrt()
        assert float((r.cpu().double() - want).abs().max() / want.abs().max()) < 1e-4


@pytest.mark.parametrize("dt", HALF)
def test_fused_qkv_rope_with_folded_layernorm(dev, dt):
    """GAR_EPI_QKV_ROPE fed with the residual stream itself (row_scale = rstd, folded weight / bias): q, k, v equal the fp64
    LayerNorm -> qkv linear -> interleaved RoPE (-> q scale) of the same bf16 x, in attention layout."""
    from gar_amd import ops
    T, n, npt, hd, H = 9, 1024, 1, 64, 16
    N, D, Kd = n + npt, 16 * 64, 256
    Npad = (N + 63) // 64 * 64
    x = q(rnd(T * N, Kd, seed=110) * 1.3 + 0.2, dt)
    W = rnd(3 * D, Kd, seed=111, scale=Kd ** -0.5)
    gamma, beta, b = 1.0 + 0.1 * rnd(Kd, seed=112), 0.1 * rnd(Kd, seed=113), 0.2 * rnd(3 * D, seed=114)
    Wc, bp = _fold_ln(W, b, gamma, beta)
    Wf, bf = q(Wc, dt), q(bp, dt)
    sin = torch.sin(rnd(n, hd // 2, seed=115)).repeat_interleave(2, -1).contiguous().to(dev)
    cos = torch.cos(rnd(n, hd // 2, seed=115)).repeat_interleave(2, -1).contiguous().to(dev)
    qs = hd ** -0.5 * 1.4426950408889634
    xd = x.to(dev, dt)
    rstd = torch.empty(T * N, dtype=torch.float32, device=dev)
    ops.row_rstd(xd, 1e-5, False, rstd)
    Q1, K1, V1 = (torch.zeros(T, H, Npad, hd, dtype=dt, device=dev) for _ in range(3))
    dummy = torch.empty(T * N, D, dtype=dt, device=dev)
    assert ops.gemm_qkv_rope(xd, Wf.to(dev, dt), bf.to(dev, dt), dummy, Q1, K1, sin, cos, H, hd, N, Npad, npt, qs, V=V1,
                             row_scale=rstd)
    xs = x.double()
    full = ((xs @ Wf.double().T) * (xs.var(1, unbiased=False, keepdim=True) + 1e-5).rsqrt() + bf.double()).view(T, N, 3, H, hd)
    qr, kr, vr = full[:, :, 0], full[:, :, 1], full[:, :, 2]

    def rope(t):
        tr = t.clone()
        body = t[:, npt:]
        rot = torch.stack([-body[..., 1::2], body[..., 0::2]], -1).reshape(body.shape)
        tr[:, npt:] = body * cos.cpu().double()[None, :, None, :] + rot * sin.cpu().double()[None, :, None, :]
        return tr
    close(Q1[:, :, :N], (rope(qr) * qs).permute(0, 2, 1, 3), dt, extra=1.5)
    close(K1[:, :, :N], rope(kr).permute(0, 2, 1, 3), dt, extra=1.5)
    close(V1[:, :, :N], vr.permute(0, 2, 1, 3), dt, extra=1.5)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("hd,Hq,Hkv,B,S", [(64, 8, 2, 3, 3700), (128, 4, 2, 3, 2800), (64, 8, 2, 130, 90)])
@pytest.mark.parametrize("fold", [False, True])
@pytest.mark.parametrize("padded", [False, True])
def test_fused_llm_qkv_rope_gemm_matches_gemm_plus_qkv_post(dev, hd, Hq, Hkv, B, S, fold, padded, dt):
    """GAR_EPI_QKV_ROPE_LLM (half-split RoPE, q scale and the KV-cache append in the qkv GEMM's epilogue; W rows in
    llm_qkv_weight_order) against the two-kernel path (GAR_EPI_NONE GEMM -> gar_llm_qkv_post, natural W order) and against an
    fp64 statement of HF's apply_rotary_pos_emb; the fused path rounds to bf16 once instead of twice. With a left-padded
    batch (RoPE position = row - left_pad), a start position > 0, a folded RMSNorm (row_scale), and sequences shorter than a
    wave's 128-row strip (S = 90: the per-row division path)."""
    from gar_amd import ops
    from oracle import gar_oracle as O
    Kd = 256
    Wd = (Hq + 2 * Hkv) * hd
    p0 = 5
    Smax = (p0 + S + 63) // 64 * 64
    Spad = (S + 63) // 64 * 64
    pos = torch.arange(Smax, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = pos[:, None] * inv[None]
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    qs = hd ** -0.5 * 1.4426950408889634
    a = q(rnd(B * S, Kd, seed=120) * 1.2, dt)
    W = q(rnd(Wd, Kd, seed=121, scale=Kd ** -0.5), dt)
    ad, Wn = a.to(dev, dt), W.to(dev, dt)
    Wp = W[ops.llm_qkv_weight_order(hd, Hq, Hkv)].contiguous().to(dev, dt)
    pads = [(7 * b) % max(S - 1, 1) for b in range(B)] if padded else None
    lp = torch.tensor(pads, dtype=torch.int32, device=dev) if padded else None
    rstd = None
    if fold:
        rstd = torch.empty(B * S, dtype=torch.float32, device=dev)
        ops.row_rstd(ad, 1e-5, True, rstd)
    Q1 = torch.zeros(B, Hq, Spad, hd, dtype=dt, device=dev)
    K1, V1 = (torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev) for _ in range(2))
    assert ops.gemm_qkv_rope_llm(ad, Wp, Q1, K1, V1, cos.to(dev), sin.to(dev), B, S, Spad, Hq, Hkv, hd, Smax, p0, None, qs,
                                 left_pad=lp, row_scale=rstd)
    # two-kernel path
    qkv = torch.empty(B * S, Wd, dtype=dt, device=dev)
    ops.gemm(ad, Wn, qkv, row_scale=rstd)
    Q2 = torch.zeros_like(Q1)
    K2, V2 = torch.zeros_like(K1), torch.zeros_like(V1)
    ops.llm_qkv_post(qkv, cos.to(dev), sin.to(dev), Q2, K2, V2, B, S, Spad, Hq, Hkv, hd, Smax, p0, None, qs, left_pad=lp)
    close(Q1, Q2, dt, extra=2.0)
    close(K1, K2, dt, extra=2.0)
    close(V1, V2, dt, extra=2.0)
    assert float(K1[:, :, :p0].float().abs().max()) == 0.0 and float(K1[:, :, p0 + S:].float().abs().max()) == 0.0
    assert float(V1[:, :, :p0].float().abs().max()) == 0.0 and float(Q1[:, :, S:].float().abs().max()) == 0.0
    # fp64 statement
    x = a.double() @ W.double().T
    if fold:
        x = x * (a.double().pow(2).mean(1, keepdim=True) + 1e-5).rsqrt()
    x = x.view(B, S, Wd)
    xq = x[..., :Hq * hd].view(B, S, Hq, hd).transpose(1, 2)
    xk = x[..., Hq * hd:(Hq + Hkv) * hd].view(B, S, Hkv, hd).transpose(1, 2)
    xv = x[..., (Hq + Hkv) * hd:].view(B, S, Hkv, hd).transpose(1, 2)
    rp = torch.stack([(p0 + torch.arange(S) - (pads[b] if padded else 0)).clamp(min=0) for b in range(B)])     # [B, S]
    c = torch.cat([cos, cos], -1).double()[rp][:, None]
    s_ = torch.cat([sin, sin], -1).double()[rp][:, None]
    rq = (xq * c + O._rotate_half(xq) * s_) * qs
    rk = xk * c + O._rotate_half(xk) * s_
    close(Q1[:, :, :S], rq, dt, extra=1.5)
    close(K1[:, :, p0:p0 + S], rk, dt, extra=1.5)
    close(V1[:, :, p0:p0 + S], xv, dt, extra=1.5)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd", [64, 128])
def test_llm_qkv_post_and_decode_attention_take_strip_ordered_heads(dev, dt, hd):
    """ABI v10: bf16 keeps ONE copy of the Llama qkv weight — folded, rows in GAR_EPI_QKV_ROPE_LLM's strip order — so the
    un-fused consumers of its product (gar_llm_qkv_post for shapes the fused epilogue does not take, gar_attention_decode_qkv in
    every decode step) read q / k head columns in that order: same outputs, bit for bit, as the natural order."""
    from gar_amd import ops
    B, S, Hq, Hkv = 3, 70, 4, 2
    Wd = (Hq + 2 * Hkv) * hd
    Smax, Spad = 128, 128
    order = ops.llm_qkv_weight_order(hd, Hq, Hkv)
    assert bool((order != torch.arange(Wd)).any()) == (hd == 128)
    pos = torch.arange(Smax, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = (pos[:, None] * inv[None]).to(dev)
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    qs = hd ** -0.5 * 1.4426950408889634
    x = q(rnd(B * S, Wd, seed=140), dt)
    xn, xp = x.to(dev, dt), x[:, order].contiguous().to(dev, dt)
    lp = torch.tensor([0, 5, 11], dtype=torch.int32, device=dev)
    outs = []
    for src, strip in ((xn, False), (xp, True)):
        Q = torch.zeros(B, Hq, Spad, hd, dtype=dt, device=dev)
        K, V = (torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev) for _ in range(2))
        ops.llm_qkv_post(src, cos, sin, Q, K, V, B, S, Spad, Hq, Hkv, hd, Smax, 0, None, qs, left_pad=lp, strip_order=strip)
        outs.append((Q, K, V))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    # one decode step on top of those caches: two-call form and the fused launch, natural vs strip order
    x1 = q(rnd(B, Wd, seed=141), dt)
    counters = torch.tensor([S, S + 1], dtype=torch.int32, device=dev)
    res = []
    for src, strip in ((x1.to(dev, dt), False), (x1[:, order].contiguous().to(dev, dt), True)):
        Q1 = torch.empty(B, Hq, 1, hd, dtype=dt, device=dev)
        K, V = outs[0][1].clone(), outs[0][2].clone()
        ops.llm_qkv_post(src, cos, sin, Q1, K, V, B, 1, 1, Hq, Hkv, hd, Smax, 0, counters[0:1], qs, left_pad=lp, strip_order=strip)
        ws = torch.empty(ops.attention_decode_workspace(B, Hq, hd, 2), dtype=torch.uint8, device=dev)
        o2 = torch.full((B, Hq * hd), float("nan"), dtype=dt, device=dev)
        ops.attention_decode(Q1, K, V, o2, B, Hq, Hkv, hd, Smax, counters[1:2], 2, ws, kv_start=lp)
        Kf, Vf, o3 = outs[0][1].clone(), outs[0][2].clone(), torch.full_like(o2, float("nan"))
        took = ops.attention_decode_qkv(src, cos, sin, Kf, Vf, o3, B, Hq, Hkv, hd, Smax, counters[0:1], qs, 2, ws, left_pad=lp,
                                        strip_order=strip)
        assert took == (dt in HALF)
        if took:
            assert torch.equal(o3, o2) and torch.equal(Kf, K) and torch.equal(Vf, V)
        res.append((Q1, K, V, o2))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("dt", DT)
@pytest.mark.parametrize("hd", [64, 128])
@pytest.mark.parametrize("S", [70, 257])
def test_decode_attention_reads_the_last_prompt_row_in_place(dev, dt, hd, S):
    """gar_attention_decode(q_stride=): the query of (b, head) is row S - 1 of a prefill's Q [B, Hq, Spad, hd], read where the
    qkv GEMM left it (the pruned last prefill layer) — equal, bit for bit, to the packed [B, Hq, hd] copy of those rows, and to
    row S - 1 of the causal prefill attention within the dtype's tolerance. Left-padded rows included."""
    from gar_amd import ops
    B, Hq, Hkv = 3, 4, 2
    Wd = (Hq + 2 * Hkv) * hd
    Smax = (S + 8 + 63) // 64 * 64
    Spad = (S + 63) // 64 * 64
    pos = torch.arange(Smax, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = (pos[:, None] * inv[None]).to(dev)
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    qs = hd ** -0.5 * 1.4426950408889634
    qkv = q(rnd(B * S, Wd, seed=150), dt).to(dev, dt)
    lp = torch.tensor([0, 9, 40], dtype=torch.int32, device=dev)
    Q = torch.zeros(B, Hq, Spad, hd, dtype=dt, device=dev)
    Kc, Vc = (torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev) for _ in range(2))
    ops.llm_qkv_post(qkv, cos, sin, Q, Kc, Vc, B, S, Spad, Hq, Hkv, hd, Smax, 0, None, qs, left_pad=lp)
    full = torch.empty(B * S, Hq * hd, dtype=dt, device=dev)
    ops.attention(Q, Kc, Vc, full, B, Hq, Hkv, hd, S, Spad, S, Smax, causal=True, kv_start=lp, v_row_major=True)
    kvl = torch.tensor([S], dtype=torch.int32, device=dev)
    for nsplit in (1, 4):
        ws = torch.empty(ops.attention_decode_workspace(B, Hq, hd, nsplit), dtype=torch.uint8, device=dev)
        o_in = torch.full((B, Hq * hd), float("nan"), dtype=dt, device=dev)
        ops.attention_decode(Q[:, :, S - 1], Kc, Vc, o_in, B, Hq, Hkv, hd, Smax, kvl, nsplit, ws, kv_start=lp, q_stride=Spad * hd)
        o_pk = torch.full_like(o_in, float("nan"))
        ops.attention_decode(Q[:, :, S - 1].contiguous(), Kc, Vc, o_pk, B, Hq, Hkv, hd, Smax, kvl, nsplit, ws, kv_start=lp)
        assert torch.equal(o_in, o_pk)
        close(o_in, full.view(B, S, Hq * hd)[:, S - 1].float().cpu(), dt, extra=2.0)


def test_tokens_add(dev):
    """gar_tokens_add: x[t, off + p, :] += add[t, p, :] (the reference's `x + mask_embeds.flatten(2).transpose(1, 2)`)."""
    from gar_amd import ops
    for dt in DT:
        T, tin, off, D = 3, 37, 1, 64
        x0 = q(rnd(T, tin + off, D, seed=160), dt)
        add = q(rnd(T, tin, D, seed=161), dt)
        x = x0.to(dev, dt).clone()
        ops.tokens_add(x, add.to(dev, dt), off)
        want = x0.clone()
        want[:, off:] = q(x0[:, off:] + add, dt)
        assert torch.equal(x.float().cpu(), want)


@pytest.mark.parametrize("dt", HALF)
def test_fused_llm_qkv_rope_start_position_from_device_memory(dev, dt):
    """qkv_pos_dev overrides qkv_pos0 (graph replay convention of gar_llm_qkv_post)."""
    from gar_amd import ops
    hd, Hq, Hkv, B, S, Kd = 64, 8, 2, 3, 3700, 128
    Smax, Spad = 3840, 3712
    pos = torch.arange(Smax, dtype=torch.float32)
    inv = 1.0 / (10000.0 ** (torch.arange(0, hd, 2).float() / hd))
    fr = (pos[:, None] * inv[None]).to(dev)
    cos, sin = fr.cos().contiguous(), fr.sin().contiguous()
    a = q(rnd(B * S, Kd, seed=130), dt).to(dev, dt)
    W = q(rnd((Hq + 2 * Hkv) * hd, Kd, seed=131, scale=Kd ** -0.5), dt).to(dev, dt)
    outs = []
    for dev_pos in (False, True):
        Q1 = torch.zeros(B, Hq, Spad, hd, dtype=dt, device=dev)
        K1, V1 = (torch.zeros(B, Hkv, Smax, hd, dtype=dt, device=dev) for _ in range(2))
        pd = torch.tensor([9], dtype=torch.int32, device=dev) if dev_pos else None
        assert ops.gemm_qkv_rope_llm(a, W, Q1, K1, V1, cos, sin, B, S, Spad, Hq, Hkv, hd, Smax, 0 if dev_pos else 9, pd, 0.18)
        outs.append((Q1, K1, V1))
    for x, y in zip(*outs):
        assert torch.equal(x, y)


@pytest.mark.parametrize("dt", HALF)
def test_fused_llm_qkv_rope_refuses_small_problems(dev, dt):
    """below the tile GEMM's size the fused epilogue does not exist: False (nothing launched), the caller keeps two kernels."""
    from gar_amd import ops
    hd, Hq, Hkv, B, S, Kd = 64, 4, 2, 2, 100, 128
    z = lambda *sh: torch.zeros(*sh, dtype=dt, device=dev)
    t = torch.zeros(128, hd // 2, device=dev)
    assert not ops.gemm_qkv_rope_llm(z(B * S, Kd), z((Hq + 2 * Hkv) * hd, Kd), z(B, Hq, 128, hd), z(B, Hkv, 128, hd),
                                     z(B, Hkv, 128, hd), t, t, B, S, 128, Hq, Hkv, hd, 128, 0, None, 0.18)


def test_abi_errors_are_reported_not_thrown(dev):
    from gar_amd import hip, ops
    a = torch.zeros(4, 60, device=dev)
    w = torch.zeros(16, 60, device=dev)
    with pytest.raises(hip.GarError, match="K=60"):
        ops.gemm(a, w, torch.zeros(4, 16, device=dev))


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("npt,hd,H", [(1, 64, 16), (0, 128, 8), (0, 96, 16)])
@pytest.mark.parametrize("compact", [True, False])
def test_fused_qkv_rope_gemm_matches_gemm_plus_qkv_post(dev, npt, hd, H, compact, dt):
    """GAR_EPI_QKV_ROPE (q / k rotated, scaled and laid out by the qkv GEMM's epilogue + gar_vit_v_transpose) against the
    two-kernel path (GAR_EPI_BIAS GEMM -> gar_vit_qkv_post) and against an fp64 statement; the fused path rounds to
    bf16 once instead of twice, so the comparison is within bf16 rounding, not bitwise. Pad rows stay zero."""
    from gar_amd import hip, ops
    # compact: (sin, cos)-pair table + barrier-free per-wave epilogue; not compact: full tables + workgroup-level epilogue
    T, n = 4, 1024
    N = n + npt
    D = H * hd
    Npad = (N + 63) // 64 * 64
    a = q(rnd(T * N, 256, seed=70), dt).to(dev, dt)
    w = q(rnd(3 * D, 256, seed=71, scale=256 ** -0.5), dt).to(dev, dt)
    b = q(rnd(3 * D, seed=72), dt).to(dev, dt)
    sin = torch.sin(rnd(n, hd // 2, seed=73)).repeat_interleave(2, -1).contiguous().to(dev)
    cos = torch.cos(rnd(n, hd // 2, seed=73)).repeat_interleave(2, -1).contiguous().to(dev)
    qs = hd ** -0.5 * 1.4426950408889634
    qkv = torch.empty(T * N, 3 * D, dtype=dt, device=dev)
    Q0, K0 = (torch.zeros(T, H, Npad, hd, dtype=dt, device=dev) for _ in range(2))
    V0 = torch.empty(T, H, hd, Npad, dtype=dt, device=dev)
    ops.gemm(a, w, qkv, hip.EPI_BIAS, bias=b)
    ops.vit_qkv_post(qkv, sin, cos, Q0, K0, V0, T, N, npt, H, hd, Npad, qs)
    Q1, K1 = (torch.zeros(T, H, Npad, hd, dtype=dt, device=dev) for _ in range(2))
    V1 = torch.full((T, H, hd, Npad), 7.0, dtype=dt, device=dev)
    vrow = torch.empty(T * N, D, dtype=dt, device=dev)
    assert ops.gemm_qkv_rope(a, w, b, vrow, Q1, K1, sin, cos, H, hd, N, Npad, npt, qs, compact=compact)
    ops.vit_v_transpose(vrow, V1, T, N, H, hd, Npad)
    # fp64 statement
    full = (a.double() @ w.double().T + b.double()).view(T, N, 3, H, hd).cpu()
    qr, kr, vr = full[:, :, 0], full[:, :, 1], full[:, :, 2]

    def rope(x):
        xr = x.clone()
        body = x[:, npt:]
        rot = torch.stack([-body[..., 1::2], body[..., 0::2]], -1).reshape(body.shape)
        xr[:, npt:] = body * cos.cpu().double()[None, :, None, :] + rot * sin.cpu().double()[None, :, None, :]
        return xr
    qref = (rope(qr) * qs).permute(0, 2, 1, 3)
    kref = rope(kr).permute(0, 2, 1, 3)
    close(Q1[:, :, :N], qref, dt)
    close(K1[:, :, :N], kref, dt)
    close(V1[:, :, :, :N], vr.permute(0, 2, 3, 1), dt)
    close(Q1, Q0.double().cpu(), dt)
    close(K1, K0.double().cpu(), dt)
    assert torch.equal(V1, V0)                              # v takes no arithmetic after the bias: identical
    if Npad > N:
        assert float(Q1[:, :, N:].abs().max()) == 0 and float(K1[:, :, N:].abs().max()) == 0
        assert float(V1[:, :, :, N:].abs().max()) == 0
    # v written head-major by the epilogue itself (gar_gemm_params.qkv_v): the same values, [T, H, Npad, hd]; the
    # row-major output is then left alone
    Q2, K2 = (torch.zeros(T, H, Npad, hd, dtype=dt, device=dev) for _ in range(2))
    V2 = torch.zeros(T, H, Npad, hd, dtype=dt, device=dev)
    vrow2 = torch.full((T * N, D), 3.0, dtype=dt, device=dev)
    assert ops.gemm_qkv_rope(a, w, b, vrow2, Q2, K2, sin, cos, H, hd, N, Npad, npt, qs, compact=compact, V=V2)
    if compact:
        # the compact table goes with the branch-free per-wave epilogue (only with V=): the q scale is folded into
        # (sin, cos) before the rotation there, after it in the general epilogue Q1 came from — fp32 rounding order only
        close(Q2[:, :, :N], qref, dt)
        close(K2[:, :, :N], kref, dt)
        close(Q2, Q1.double().cpu(), dt)
        close(K2, K1.double().cpu(), dt)
    else:
        assert torch.equal(Q2, Q1) and torch.equal(K2, K1)
    assert torch.equal(V2[:, :, :N], V1.transpose(2, 3)[:, :, :N])
    assert float((vrow2 - 3.0).abs().max()) == 0 and (Npad == N or float(V2[:, :, N:].abs().max()) == 0)
