#!/usr/bin/env python
"""Micro-benchmark of the HBM-bound replay pass kernels at GAR-1B sizes (P=16, C=2048, 17 tiles, S=4718):
roi_replay_batched over n jobs (one crop token per region), pool2x2 and embed_assemble for n regions. HIP events."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    P, C, tiles, S, V = 16, 2048, 17, 4718, 128262
    # video replay (BASELINE configs[4]): GAR-8B width, 8 frames per clip, one 256-row replay per frame on its own P x P map
    Cv, fr = 4096, 8
    for n in (16, 64):
        feats = torch.randn(n * fr, P * P, Cv, device=dev).to(dt)
        Sv = 4296
        emb = torch.zeros(n, Sv, Cv, device=dev, dtype=dt)
        spans = torch.full((n, fr, 2), -1, dtype=torch.int32)
        for f in range(fr):
            spans[:, f, 0], spans[:, f, 1] = 100 + 512 * f, 100 + 512 * f + 255
        spans = spans.to(dev)
        jobs = ops.roi_jobs_tensor([(b, f, f, 1, 1, 3.1 + f, 5.6, 9.8 + f * 0.5, 13.1, 1.0 / 28) for b in range(n)
                                    for f in range(fr)], dev)
        t = timeit(lambda: ops.roi_replay_batched(feats, emb, spans, jobs, fr, fr, P, Cv, Sv))
        nb = n * fr * (P * P + 16) * Cv * 2
        print(f"roi_replay_batched video clips={n:3d} ({n * fr} jobs, C={Cv}): {t * 1e6:8.1f} us  {nb / 1e6:7.2f} MB  "
              f"{nb / t / 1e9:7.1f} GB/s", flush=True)
    for n in (1, 16, 64):
        feats = torch.randn(n * tiles, P * P, C, device=dev).to(dt)
        emb = torch.zeros(n, S, C, device=dev, dtype=dt)
        spans = torch.full((n, 5, 2), -1, dtype=torch.int32)
        spans[:, 1, 0], spans[:, 1, 1] = 4400, 4655
        spans = spans.to(dev)
        jobs = ops.roi_jobs_tensor([(b, 1, 1, 4, 4, 46.1, 55.6, 50.8, 59.1, 1.0 / 28) for b in range(n)], dev)
        t = timeit(lambda: ops.roi_replay_batched(feats, emb, spans, jobs, 5, tiles, P, C, S))
        nb = n * (P * P + 16) * C * 2
        print(f"roi_replay_batched n={n:3d}: {t * 1e6:8.1f} us  {nb / 1e6:7.2f} MB  {nb / t / 1e9:7.1f} GB/s", flush=True)
        x = torch.randn(n * tiles, 1025, C, device=dev).to(dt)
        y = torch.empty(n * tiles, P * P, C, device=dev, dtype=dt)
        t = timeit(lambda: ops.pool2x2(x, y, 32, in_tile_tokens=1025, in_token_offset=1))
        nb = (n * tiles * 1024 * C + y.numel()) * 2
        print(f"pool2x2            n={n:3d}: {t * 1e6:8.1f} us  {nb / 1e6:7.2f} MB  {nb / t / 1e9:7.1f} GB/s", flush=True)
        ids = torch.randint(0, V, (n, S), device=dev)
        ids[:, 10:10 + tiles * P * P] = 128002
        slot = torch.empty(n, S, dtype=torch.int32, device=dev)
        counts = torch.empty(n, dtype=torch.int32, device=dev)
        sp2 = torch.empty(n, 5, 2, dtype=torch.int32, device=dev)
        crop = torch.tensor([128004, 128005, 128008, 128010, 128011], device=dev)
        ops.placeholder_scan(ids, 128002, crop, slot, counts, sp2)
        E = torch.randn(V, C, device=dev).to(dt)
        t = timeit(lambda: ops.embed_assemble(ids, slot, E, y, emb, tiles * P * P))
        nb = 2 * emb.numel() * 2
        print(f"embed_assemble     n={n:3d}: {t * 1e6:8.1f} us  {nb / 1e6:7.2f} MB  {nb / t / 1e9:7.1f} GB/s", flush=True)


if __name__ == "__main__":
    main()
