#!/usr/bin/env python
"""Micro-benchmark of llm_qkv_post (RoPE + q scale + KV-cache write with V transposed) at the prefill shape of the
benchmark: 16 sequences x 4718 tokens, 32 q / 8 kv heads of 64."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    B, S, Hq, Hkv, hd, Smax = 16, 4718, 32, 8, 64, 4864
    Spad = (S + 63) // 64 * 64
    Wd = (Hq + 2 * Hkv) * hd
    qkv = torch.randn(B * S, Wd, device=dev).to(dt)
    cos = torch.randn(Smax, hd // 2, device=dev)
    sin = torch.randn(Smax, hd // 2, device=dev)
    Q = torch.empty(B, Hq, Spad, hd, device=dev, dtype=dt)
    Kc = torch.zeros(B, Hkv, Smax, hd, device=dev, dtype=dt)
    Vc = torch.zeros(B, Hkv, Smax, hd, device=dev, dtype=dt)

    def run():
        ops.llm_qkv_post(qkv, cos, sin, Q, Kc, Vc, B, S, Spad, Hq, Hkv, hd, Smax, 0, None, 0.18)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    nb = 2 * B * S * Wd * 2
    print(f"llm_qkv_post B={B} S={S}: {us:7.1f} us  {nb / us / 1e6:5.2f} TB/s (read + write)", flush=True)


if __name__ == "__main__":
    main()
