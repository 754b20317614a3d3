#!/usr/bin/env python
"""Micro-benchmark of LayerNorm / RMSNorm at the benchmark's shapes (HIP events)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    for name, M, D, rms in (("vit LN", 278800, 1024, False), ("prefill RMS", 75488, 2048, True), ("decode RMS", 64, 2048, True)):
        x = torch.randn(M, D, device=dev).to(dt)
        y = torch.empty_like(x)
        w = torch.randn(D, device=dev).to(dt)
        b = torch.randn(D, device=dev).to(dt)
        fn = (lambda: ops.rmsnorm(x, w, 1e-5, out=y)) if rms else (lambda: ops.layernorm(x, w, b, 1e-5, out=y))
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print(f"{name:12s} M={M:6d} D={D}: {us:8.1f} us  {2 * M * D * 2 / us / 1e6:6.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
