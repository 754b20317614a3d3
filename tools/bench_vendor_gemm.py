#!/usr/bin/env python
"""Calibration only, not a product path: what the vendor library behind torch.mm (hipBLASLt / rocBLAS) reaches on the planner's
GEMM shapes with NO epilogue, next to the hand-written tile GEMM with GAR_EPI_NONE — i.e. how far a bf16 GEMM gets on this
part under its power budget. The product never calls it (tests/test_abi.py: libgar_hip.so links no vendor math library)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402

SHAPES = [("vit qkv", 396675, 3072, 1024), ("vit proj", 396675, 1024, 1024), ("vit fc1", 396675, 4096, 1024),
          ("vit fc2", 396675, 1024, 4096), ("llm qkv", 122668, 3072, 2048), ("llm o", 122668, 2048, 2048),
          ("llm gate/up", 122668, 16384, 2048), ("llm down", 122668, 2048, 8192), ("square 8k", 8192, 8192, 8192)]


def timed(fn, reps=5):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    hip.require_device(0)
    dev = "cuda:0"
    tf = tv = fl_all = 0.0
    for name, M, N, K in SHAPES:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        ms_v = timed(lambda: torch.mm(a, w.t(), out=out))
        ms_o = timed(lambda: ops.gemm(a, w, out))
        fl = 2.0 * M * N * K
        fl_all += fl
        tv += ms_v
        tf += ms_o
        print(f"{name:12s} M={M:6d} N={N:5d} K={K:4d}  vendor {ms_v:7.3f} ms {fl / ms_v / 1e9:7.1f} TFLOP/s   "
              f"this repo (EPI_NONE) {ms_o:7.3f} ms {fl / ms_o / 1e9:7.1f} TFLOP/s", flush=True)
    print(f"weighted: vendor {fl_all / tv / 1e9:.1f} TFLOP/s, this repo {fl_all / tf / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    main()
