#!/usr/bin/env python
"""Micro-benchmark of the decode (skinny, M <= 64) bf16 GEMMs on GAR-1B's decode shapes: 32 back-to-back launches
captured in one hipGraph, HIP-event timed; reports us per launch and the weight-stream rate."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402

SHAPES = [("qkv", 3072, 2048, hip.EPI_NONE), ("o", 2048, 2048, hip.EPI_RES), ("gate/up", 16384, 2048, hip.EPI_SWIGLU),
          ("down", 2048, 8192, hip.EPI_RES), ("lm_head", 128262, 2048, hip.EPI_NONE)]


SHAPES_8B = [("qkv", 6144, 4096, hip.EPI_NONE), ("o", 4096, 4096, hip.EPI_RES), ("gate/up", 28672, 4096, hip.EPI_SWIGLU),
             ("down", 4096, 14336, hip.EPI_RES), ("lm_head", 128262, 4096, hip.EPI_NONE)]


def main():
    global SHAPES
    if os.environ.get("MODEL") == "8b":         # Llama-3.1-8B decode shapes (GAR-8B)
        SHAPES = SHAPES_8B
    hip.require_device(0)
    dev = "cuda:0"
    L = 16
    for M in (int(a) for a in (sys.argv[1:] or ["16", "64"])):
        tot = 0.0
        for name, N, K, epi in SHAPES:
            # L distinct weight matrices; COLD=1: as many as it takes to stream >= 1.5 GB per round, so that NO shape is
            # served by the 256 MB Infinity Cache (16 x qkv = 200 MB and 16 x o = 134 MB are: inside the decode step, where
            # 2.47 GB of weights pass per token, they are not)
            nl = L if N < 100000 else 2
            if os.environ.get("COLD") == "1":
                nl = max(nl, -(-1_500_000_000 // (N * K * 2)))
            ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(nl)]
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            out = torch.zeros(M, N // 2 if epi == hip.EPI_SWIGLU else N, device=dev, dtype=torch.bfloat16)
            kw = {"residual": out} if epi == hip.EPI_RES else {}

            def run():
                for w in ws:
                    ops.gemm(a, w, out, epi, **kw)
            run()
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                run()
            g.replay()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                g.replay()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 / len(ws) * 1e3
            nb = N * K * 2
            per_step = us * (16 if N < 100000 else 1)
            tot += per_step
            print(f"M={M:3d} {name:8s} N={N:6d} K={K:5d}  {us:8.1f} us  {nb / us / 1e6:7.2f} TB/s", flush=True)
        print(f"M={M:3d} GEMMs of one decode step (16 layers + head): {tot / 1e3:.3f} ms", flush=True)
        if M > 16:      # `down` as split-K partials + the fused reduce / residual / RMSNorm launch, against down(EPI_RES) + rmsnorm
            N, K = (4096, 14336) if os.environ.get("MODEL") == "8b" else (2048, 8192)
            ws = [(torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16) for _ in range(L)]
            a = torch.randn(M, K, device=dev).to(torch.bfloat16)
            h = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
            y = torch.empty_like(h)
            g = torch.ones(N, device=dev, dtype=torch.bfloat16)
            for S in ((1, 2, 4) if N > 2048 else (1, 2, 4, 8)):
                part = torch.empty(S, M, N, device=dev, dtype=torch.float32)

                def run(what):
                    for w in ws:
                        if S == 1:
                            if what != "reduce":
                                ops.gemm(a, w, h, hip.EPI_RES, residual=h)
                            if what != "gemm":
                                ops.rmsnorm(h, g, 1e-5, out=y)
                        else:
                            if what != "reduce":
                                ops.gemm(a, w, None, partial=part)
                            if what != "gemm":
                                ops.splitk_residual_rmsnorm(part, h, g, 1e-5, out=y)
                res = {}
                for what in ("both", "gemm", "reduce"):
                    run(what)
                    torch.cuda.synchronize()
                    gr = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(gr):
                        run(what)
                    gr.replay()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(5):
                        gr.replay()
                    e1.record()
                    torch.cuda.synchronize()
                    res[what] = e0.elapsed_time(e1) / 5 / len(ws) * 1e3
                print(f"M={M:3d} down + residual + next RMSNorm, split_k={S}: {res['both']:6.1f} us per layer "
                      f"(GEMM alone {res['gemm']:5.1f}, reduce / norm launch alone {res['reduce']:5.1f})", flush=True)


if __name__ == "__main__":
    main()
