#!/usr/bin/env python
"""Same-process A/B of two builds of libgar_hip.so on the tile GEMM's shapes: outputs compared BIT FOR BIT, launches timed
alternately with HIP events (the pool's boxes differ by more than most kernel changes: only same-box pairs mean anything).

    python tools/ab_lw.py [A.so] [B.so]        default: the product library against variants/libgar_hip_lwoff.so
    SHAPESET=plan|k2048|tails|all  REPS=n ROUNDS=n  CHECK=0|1  FOLD=0|1

`lwoff` = tools/build_variant.sh lwoff gemm_lw -DLW_OFF (every tile GEMM on the 8-wave ping-pong kernel)."""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402

V = os.path.join(ROOT, "grasp-any-region_amd", "gar_amd", "variants")
PLAN = [("vit qkv rope", 396675, 3072, 1024, hip.EPI_QKV_ROPE), ("vit proj", 396675, 1024, 1024, hip.EPI_BIAS_SCALE_RES),
        ("vit fc1", 396675, 4096, 1024, hip.EPI_BIAS_GELU), ("vit fc2", 396675, 1024, 4096, hip.EPI_BIAS_SCALE_RES),
        ("llm qkv rope", 122668, 3072, 2048, hip.EPI_QKV_ROPE_LLM), ("llm o", 122668, 2048, 2048, hip.EPI_RES),
        ("llm gate/up", 122668, 16384, 2048, hip.EPI_SWIGLU), ("llm down", 122668, 2048, 8192, hip.EPI_RES)]
K2048 = [s for s in PLAN if s[3] >= 2048] + [("proj2 bias", 69632, 2048, 2048, hip.EPI_BIAS), ("llm qkv none", 122668, 3072, 2048, hip.EPI_NONE)]
# M / N tails, odd K-tile counts, more tiles than CUs, every epilogue the 4-wave kernel instantiates
TAILS = [(n, M, N, K, e) for (M, N, K) in [(4099, 2048, 256), (2049, 4096, 1024), (33000, 1024, 320), (9000, 2048, 448), (16640, 2056, 2048),
                                           (70000, 1000, 4096)]
         for n, e in [("none", hip.EPI_NONE), ("bias", hip.EPI_BIAS), ("res", hip.EPI_RES), ("bsr", hip.EPI_BIAS_SCALE_RES),
                      ("swiglu", hip.EPI_SWIGLU)] if not (e == hip.EPI_SWIGLU and N % 32)]


def load(path):
    lib = C.CDLL(path)
    for name, (argtypes, restype) in hip.SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
    return lib


def main():
    pa = sys.argv[1] if len(sys.argv) > 1 else hip.LIB_PATH
    pb = sys.argv[2] if len(sys.argv) > 2 else os.path.join(V, "libgar_hip_lwoff.so")
    libs = {"A": load(pa), "B": load(pb)}
    print(f"A = {pa}\nB = {pb}")
    hip.require_device(0)
    dev = "cuda:0"
    reps, rounds = int(os.environ.get("REPS", "6")), int(os.environ.get("ROUNDS", "3"))
    check = os.environ.get("CHECK", "1") == "1"
    fold = os.environ.get("FOLD", "0") == "1"
    ss = os.environ.get("SHAPESET", "k2048")
    shapes = {"plan": PLAN, "k2048": K2048, "tails": TAILS, "all": K2048 + TAILS}[ss]
    tot = {"A": [0.0, 0.0], "B": [0.0, 0.0]}
    bad = 0
    for name, M, N, K, epi in shapes:
        g = torch.Generator(device=dev).manual_seed(M + N + K + epi)
        a = torch.randn(M, K, device=dev, generator=g).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev, generator=g) * K ** -0.5).to(torch.bfloat16)
        No = N // 2 if epi == hip.EPI_SWIGLU else N
        kw = {}
        rs = None
        if fold and epi in (hip.EPI_NONE, hip.EPI_BIAS, hip.EPI_BIAS_GELU, hip.EPI_SWIGLU, hip.EPI_QKV_ROPE, hip.EPI_QKV_ROPE_LLM):
            rs = torch.rand(M, device=dev, generator=g) + 0.5
        if epi in (hip.EPI_BIAS, hip.EPI_BIAS_GELU, hip.EPI_BIAS_SCALE_RES, hip.EPI_QKV_ROPE):
            kw["bias"] = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        if epi == hip.EPI_BIAS_SCALE_RES:
            kw["gamma"] = torch.randn(N, device=dev, generator=g).to(torch.bfloat16)
        res0 = torch.randn(M, No, device=dev, generator=g).to(torch.bfloat16) if epi in (hip.EPI_BIAS_SCALE_RES, hip.EPI_RES) else None
        outs, stats, extra = {}, {}, {}
        for k in libs:
            outs[k] = torch.full((M, No), 3.0, device=dev, dtype=torch.bfloat16)
            if res0 is not None and (fold or True):
                stats[k] = torch.zeros(M, (N + 63) // 64, 2, device=dev, dtype=torch.float32)
        if epi == hip.EPI_QKV_ROPE:
            H, hd, T = 16, 64, M // 1025
            ang = torch.randn(1024, hd // 2, device=dev, generator=g)
            sin, cos = (f(ang).repeat_interleave(2, -1).contiguous() for f in (torch.sin, torch.cos))
            for k in libs:
                extra[k] = [torch.zeros(T, H, 1088, hd, device=dev, dtype=torch.bfloat16) for _ in range(3)]
        if epi == hip.EPI_QKV_ROPE_LLM:
            Hq, Hkv, hd, S = 32, 8, 64, 4718
            B_ = M // S
            ang = torch.randn(4864, hd // 2, device=dev, generator=g)
            cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()
            for k in libs:
                extra[k] = [torch.zeros(B_, Hq, 4736, hd, device=dev, dtype=torch.bfloat16)] + \
                           [torch.zeros(B_, Hkv, 4864, hd, device=dev, dtype=torch.bfloat16) for _ in range(2)]

        def call(k, fresh=False):
            hip._lib = libs[k]
            o = outs[k]
            if epi == hip.EPI_QKV_ROPE:
                Q_, K_, V_ = extra[k]
                assert ops.gemm_qkv_rope(a, w, kw["bias"], o, Q_, K_, sin, cos, 16, 64, 1025, 1088, 1, 0.18, V=V_, row_scale=rs)
            elif epi == hip.EPI_QKV_ROPE_LLM:
                Q_, K_, V_ = extra[k]
                assert ops.gemm_qkv_rope_llm(a, w, Q_, K_, V_, cos, sin, M // 4718, 4718, 4736, 32, 8, 64, 4864, 0, None, 0.18, row_scale=rs)
            else:
                k2 = dict(kw)
                if res0 is not None:
                    if fresh:
                        o.copy_(res0)
                    k2["residual"] = o          # in place, as the model runs it
                    if fold:
                        k2["row_stats"] = stats[k]
                if rs is not None:
                    k2["row_scale"] = rs
                ops.gemm(a, w, o, epi, **k2)

        ok = True
        if check:
            for k in libs:
                call(k, fresh=True)
            torch.cuda.synchronize()
            ok = torch.equal(outs["A"], outs["B"])
            if fold and res0 is not None:
                ok = ok and torch.equal(stats["A"], stats["B"])
            for x, y in zip(extra.get("A", []), extra.get("B", [])):
                ok = ok and torch.equal(x, y)
            if not ok:
                bad += 1
                d = (outs["A"].float() - outs["B"].float()).abs()
                print(f"   MISMATCH {name}: max|A-B| = {float(d.max()):.4g}, {int((d > 0).sum())} of {d.numel()} elements, first rows "
                      f"{torch.nonzero(d.amax(1) > 0)[:6].flatten().tolist()} cols {torch.nonzero(d.amax(0) > 0)[:6].flatten().tolist()}")
        best = {"A": 1e9, "B": 1e9}
        if reps > 0:
            for _ in range(rounds):
                for k in libs:
                    call(k)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(reps):
                        call(k)
                    e1.record()
                    torch.cuda.synchronize()
                    best[k] = min(best[k], e0.elapsed_time(e1) / reps)
            fl = 2.0 * M * N * K
            for k in libs:
                tot[k][0] += fl
                tot[k][1] += best[k]
            print(f"{name:14s} M={M:6d} N={N:5d} K={K:4d}  A {best['A']:8.3f} ms {fl / best['A'] / 1e9:7.1f} TF   B {best['B']:8.3f} ms "
                  f"{fl / best['B'] / 1e9:7.1f} TF   A/B time {best['A'] / best['B']:.4f}   {'bit-identical' if ok and check else ('MISMATCH' if check else '')}",
                  flush=True)
        else:
            print(f"{name:14s} M={M:6d} N={N:5d} K={K:4d}  {'bit-identical' if ok else 'MISMATCH'}", flush=True)
    if reps > 0:
        print(f"weighted: A {tot['A'][0] / tot['A'][1] / 1e9:.1f} TFLOP/s   B {tot['B'][0] / tot['B'][1] / 1e9:.1f} TFLOP/s")
    print("mismatching shapes:", bad)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
