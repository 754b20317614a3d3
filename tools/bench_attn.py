#!/usr/bin/env python
"""Micro-benchmark of the bf16 flash-attention kernel at the benchmark's two shapes: ViT (272 tiles x 16 heads,
1025 tokens, non-causal) and Llama prefill (16 sequences x 32 q / 8 kv heads, 4718 tokens, causal)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    shapes = [("vit", 272, 16, 16, 64, 1025, False), ("prefill", 16, 32, 8, 64, 4718, True)]
    if os.environ.get("SHAPESET") == "all":      # + GAR-8B: PE-G/14 (head_dim 96, no cls token) and Llama-3.1-8B prefill (head_dim 128)
        shapes += [("vit 8b", 80, 16, 16, 96, 1024, False), ("prefill 8b", 16, 32, 8, 128, 1590, True)]
    for name, B, Hq, Hkv, hd, n, causal in shapes:
        npad = (n + 63) // 64 * 64
        Q = torch.randn(B, Hq, npad, hd, device=dev).to(dt) * 0.2
        K = torch.randn(B, Hkv, npad, hd, device=dev).to(dt)
        Vt = torch.randn(B, Hkv, hd, npad, device=dev).to(dt)
        O = torch.empty(B * n, Hq * hd, device=dev, dtype=dt)

        vrow = os.environ.get("VROW") == "1"
        Vr = Vt.transpose(2, 3).contiguous() if vrow else None
        pfx = int(os.environ.get("KV_PREFIX", "0")) if (vrow and not causal and n % 64 == 1) else 0    # cls key folded

        def run():
            if vrow:
                ops.attention(Q, K, Vr, O, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal, v_row_major=True, kv_prefix=pfx)
            else:
                ops.attention(Q, K, Vt, O, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        fl = 4.0 * B * Hq * n * n * hd * (0.5 if causal else 1.0)
        print(f"{name:10s} {ms:7.3f} ms  {fl / ms / 1e9:7.1f} TFLOP/s", flush=True)


if __name__ == "__main__":
    main()
