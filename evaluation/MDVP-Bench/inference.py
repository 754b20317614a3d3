#!/usr/bin/env python
"""MDVP-Bench inference on the MI355X-native path — counterpart of the reference's evaluation/MDVP-Bench/inference.py
(same flags; single GPU, or `python -m torch.distributed.run --nproc-per-node N` to shard items over N GPUs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from gar_amd.bench_loops import run_mdvp_bench  # noqa: E402

if __name__ == "__main__":
    run_mdvp_bench()
