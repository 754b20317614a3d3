"""GPU (-m gpu): the opt-in 4-wave persistent attention kernel attn_bf16_v4 (csrc/attention_v4.hip, gar_attention_v4_enable) against
the fp64 reference of the same 16-bit inputs and against the default kernel (v2) — the parity gates of the default kernel's own
tests (tests/test_gpu_ops.py: ViT tile with the folded cls key, causal GQA prefill, left-padded batch, lazy-max score profiles).
Semantics: timm Eva SDPA (modeling_perception_lm.py:210-214), flash-attn-2 causal GQA (modeling_gar.py:40-43)."""
import pytest
import torch

pytestmark = pytest.mark.gpu
HALF = [torch.bfloat16, torch.float16]
LN2 = 0.6931471805599453


@pytest.fixture()
def v4():
    from gar_amd import hip, ops
    hip.require_device(0)
    prev = ops.attention_v4_enable(1)
    yield torch.device("cuda:0")
    ops.attention_v4_enable(prev)


def _tol(dt):
    return 1.6e-2 if dt == torch.bfloat16 else 2e-3


def _close(out, ref, dt, extra=1.0):
    err = float((out.double().cpu() - ref).abs().max())
    assert err <= extra * _tol(dt) * max(float(ref.abs().max()), 1e-6) + 1e-6, err


def _ref(Q, K, V, n, kv, causal, kv_lo=None):
    Hq, Hkv = Q.shape[1], K.shape[1]
    q = Q[:, :, :n].double().cpu()
    k = K[:, :, :kv].double().cpu().repeat_interleave(Hq // Hkv, 1)
    v = V[:, :, :kv].double().cpu().repeat_interleave(Hq // Hkv, 1)
    s = q @ k.transpose(-1, -2) * LN2
    if causal:
        s = s.masked_fill(~torch.ones(n, kv, dtype=torch.bool).tril(kv - n), float("-inf"))
    if kv_lo is not None:
        for b, lo in enumerate(kv_lo):
            s[b, :, :, :lo] = float("-inf")
    p = torch.softmax(s, -1)
    p = torch.nan_to_num(p)           # rows in front of a left pad see nothing: never read
    return (p @ v).permute(0, 2, 1, 3).reshape(Q.shape[0] * n, Hq * 64)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("B,Hq,Hkv,n,causal,pfx", [(3, 2, 2, 1025, False, 1), (2, 4, 4, 1024, False, 0), (2, 8, 2, 333, True, 0),
                                                   (1, 4, 1, 1300, True, 0), (5, 2, 1, 256, True, 0), (2, 2, 2, 520, False, 0)])
def test_v4_matches_fp64_reference_and_v2(v4, dt, B, Hq, Hkv, n, causal, pfx):
    from gar_amd import ops
    dev = v4
    npad = (n + 63) // 64 * 64
    g = torch.Generator().manual_seed(5)
    Q = (torch.randn(B, Hq, npad, 64, generator=g) * 0.3).to(dt).to(dev)
    K = torch.randn(B, Hkv, npad, 64, generator=g).to(dt).to(dev)
    V = torch.randn(B, Hkv, npad, 64, generator=g).to(dt).to(dev)
    out = torch.full((B * n, Hq * 64), float("nan"), dtype=dt, device=dev)
    ops.attention(Q, K, V, out, B, Hq, Hkv, 64, n, npad, n, npad, causal=causal, v_row_major=True, kv_prefix=pfx)
    assert torch.isfinite(out.float()).all()
    _close(out, _ref(Q, K, V, n, n, causal), dt)
    ops.attention_v4_enable(0)
    out2 = torch.empty_like(out)
    ops.attention(Q, K, V, out2, B, Hq, Hkv, 64, n, npad, n, npad, causal=causal, v_row_major=True, kv_prefix=pfx)
    ops.attention_v4_enable(1)
    # two roundings of one function (row sums are accumulated in another order): far inside the tolerance against fp64
    _close(out, out2.double().cpu(), dt, extra=0.5)


@pytest.mark.parametrize("dt", HALF)
def test_v4_left_padded_causal_batch(v4, dt):
    """kv_start (left-padded batch, HF generation): keys in front of a sequence's first real position stay hidden; prompt rows of
    sequences with different pads, kv_len > q_len (a prefix already in the cache)."""
    from gar_amd import ops
    dev = v4
    B, Hq, Hkv, n, kv = 3, 4, 2, 400, 464
    npad, kvpad = 448, 512
    g = torch.Generator().manual_seed(6)
    Q = (torch.randn(B, Hq, npad, 64, generator=g) * 0.3).to(dt).to(dev)
    K = torch.randn(B, Hkv, kvpad, 64, generator=g).to(dt).to(dev)
    V = torch.randn(B, Hkv, kvpad, 64, generator=g).to(dt).to(dev)
    lo = [0, 70, 301]
    out = torch.empty(B * n, Hq * 64, dtype=dt, device=dev)
    ops.attention(Q, K, V, out, B, Hq, Hkv, 64, n, npad, kv, kvpad, causal=True, v_row_major=True,
                  kv_start=torch.tensor(lo, dtype=torch.int32, device=dev))
    ref = _ref(Q, K, V, n, kv, True, lo)
    o = out.double().cpu().view(B, n, -1)
    r = ref.view(B, n, -1)
    for b in range(B):
        first = max(0, lo[b] - (kv - n))          # query rows whose own position is in front of the pad are never read
        _close(o[b, first:], r[b, first:], dt)


@pytest.mark.parametrize("dt", HALF)
@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("profile", ["rising", "spike", "falling", "huge"])
def test_v4_lazy_max_redo_path(v4, causal, profile, dt):
    """the score profiles of test_attention_lazy_max_redo_path (tests/test_gpu_ops.py) through v4's lazy running max: a row sum that
    reaches the limit sends THAT q-block through the exact re-base from its intact scores."""
    from gar_amd import ops
    dev = v4
    B, H, n, hd = 1, 2, 64 * 9, 64
    g = torch.Generator().manual_seed(77)
    Qf = torch.randn(B, H, n, hd, generator=g)
    Kf = torch.randn(B, H, n, hd, generator=g)
    Vf = torch.randn(B, H, n, hd, generator=g)
    tile = torch.arange(n) // 64
    scale = {"rising": 1.0 + 8.0 * tile.float(), "spike": torch.where(tile == 4, 70.0, 1.0) * torch.ones(n),
             "falling": 30.0 / (1.0 + 4.0 * tile.float()), "huge": torch.where(tile >= 5, 400.0, 1.0) * torch.ones(n)}[profile]
    u = torch.nn.functional.normalize(torch.randn(hd, generator=g), dim=0)
    Qf = Qf + 4.0 * u
    Kf = Kf * 0.1 + scale[None, None, :, None] * u
    Q = (Qf * hd ** -0.5 * 1.4426950408889634).to(dt).to(dev)
    K = Kf.to(dt).to(dev)
    V = Vf.to(dt).to(dev)
    out = torch.empty(B * n, H * hd, dtype=dt, device=dev)
    ops.attention(Q, K, V, out, B, H, H, hd, n, n, n, n, causal=causal, v_row_major=True)
    assert torch.isfinite(out.float()).all()
    _close(out, _ref(Q, K, V, n, n, causal), dt, extra=2.0)
    if not causal:
        for k0_scale in (1.0, 60.0, -60.0):
            K2 = K.clone()
            K2[:, :, 0] = (K[:, :, 0].float() * k0_scale).to(dt)
            out3 = torch.full_like(out, float("nan"))
            ops.attention(Q, K2, V, out3, B, H, H, hd, n, n, n, n, causal=False, v_row_major=True, kv_prefix=1)
            # n = 576 rows with one prefix key folded: 575 = 2 x 256 + 63 query rows ... the head rows go to v2, whole blocks to v4
            assert torch.isfinite(out3.float()).all()
            _close(out3, _ref(Q, K2, V, n, n, False), dt, extra=2.0)
