#!/usr/bin/env python
"""VideoRefer-style video region captioning on the MI355X-native path (the reference has the model-side video replay,
modeling_perception_lm.py:765-852, but no caller; annotation layout in gar_amd.bench_loops.run_video_refer)
(same flags; single GPU, or `python -m torch.distributed.run --nproc-per-node N` to shard items over N GPUs)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

from gar_amd.bench_loops import run_video_refer  # noqa: E402

if __name__ == "__main__":
    run_video_refer()
