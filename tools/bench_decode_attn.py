#!/usr/bin/env python
"""Micro-benchmark of the split-KV decode attention at GAR-1B decode shapes (Hq 32, Hkv 8, hd 64, kv ~4.75k)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    Hq, Hkv, Smax, kv = 32, 8, 4800, 4750
    hd = int(os.environ.get("HD", "64"))    # 64: GAR-1B, 128: GAR-8B
    L = 4                                   # distinct caches so the stream comes from HBM
    batches = [int(x) for x in os.environ.get("BATCHES", "16,64").split(",")]
    splits = [int(x) for x in os.environ.get("NSPLITS", "1,2,4").split(",")]
    if os.environ.get("L"):
        L = int(os.environ["L"])
    for B in batches:
        Kc = [torch.randn(B, Hkv, Smax, hd, device=dev).to(dt) for _ in range(L)]
        Vc = [torch.randn(B, Hkv, Smax, hd, device=dev).to(dt) for _ in range(L)]       # same bytes in either V layout
        q = torch.randn(B, Hq, hd, device=dev).to(dt)
        O = torch.empty(B, Hq * hd, device=dev, dtype=dt)
        kvl = torch.tensor([kv], dtype=torch.int32, device=dev)
        for ns in splits:
            ws = torch.empty(ops.attention_decode_workspace(B, Hq, hd, ns), dtype=torch.uint8, device=dev)

            def run():
                for i in range(L):
                    ops.attention_decode(q, Kc[i], Vc[i], O, B, Hq, Hkv, hd, Smax, kvl, ns, ws)
            run()
            torch.cuda.synchronize()
            if os.environ.get("GRAPH") == "1":      # replayed from a hipGraph, as the decode step runs them
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g):
                    run()
                g.replay()
                launch = g.replay
            else:
                launch = run
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                launch()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 5 / L * 1e3
            nb = B * Hkv * kv * hd * 2 * 2
            print(f"B={B:3d} nsplit={ns}: {us:8.1f} us  {nb / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
