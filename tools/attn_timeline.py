#!/usr/bin/env python
"""Segment timeline of attn_bf16_v2_kernel (diagnostic build -DATTN_TIMELINE, see tools/build_variant.sh):

    tools/build_variant.sh attntl attention_bf16 -DATTN_TIMELINE
    GAR_HIP_LIB=grasp-any-region_amd/gar_amd/variants/libgar_hip_attntl.so python tools/attn_timeline.py

The (b = 0, head = 0, middle q-block) workgroup stamps s_memtime between the segments of its kv loop and writes the
per-wave sums over full, unmasked kv tiles into rows 0..3 of O. One kv tile of one wave = 8 + 8 v_mfma_f32_32x32x16_bf16
= 512 matrix-pipe cycles per SIMD; with W waves resident per SIMD the pipe is saturated when a wave's tile takes 512 W
cycles. The stamps themselves (s_memtime + lgkmcnt(0) + scheduling barriers) cost ~10 % and keep LDS reads from being
hoisted across segments."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402

SEG = ["issue next tile's DMA", "K frag reads + QK^T MFMAs (issue)", "max (waits for the QK^T results) + half exchange",
       "rescale check, exp2, row sum, bf16 pack", "Vt frag reads + PV MFMAs (issue)", "vmcnt(0): next tile landed",
       "s_barrier"]


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    for name, B, Hq, Hkv, hd, n, causal in (("vit", 272, 16, 16, 64, 1025, False), ("prefill", 16, 32, 8, 64, 4718, True)):
        npad = (n + 63) // 64 * 64
        Q = torch.randn(B, Hq, npad, hd, device=dev).to(dt) * 0.2
        K = torch.randn(B, Hkv, npad, hd, device=dev).to(dt)
        Vt = torch.randn(B, Hkv, hd, npad, device=dev).to(dt)
        O = torch.zeros(B * n, Hq * hd, device=dev, dtype=dt)
        for _ in range(3):
            ops.attention(Q, K, Vt, O, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        ops.attention(Q, K, Vt, O, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = 4.0 * B * Hq * n * n * hd * (0.5 if causal else 1.0)
        raw = O[:4].contiguous().view(torch.int32)[:, :16].cpu().to(torch.int64) & 0xffffffff
        print(f"--- {name}: B={B} Hq={Hq} Hkv={Hkv} n={n} causal={causal}; instrumented kernel {ms:.3f} ms "
              f"({fl / ms / 1e9:.0f} TFLOP/s)")
        for w in range(4):
            r = raw[w].tolist()
            ntl, total, ntiles = r[7], r[8], r[9]
            if ntl == 0:
                print(f"wave {w}: no full tiles recorded {r}")
                continue
            per = [x / ntl for x in r[:7]]
            tot = sum(per)
            print(f"wave {w}: {ntl} full tiles of {ntiles}; {tot:7.0f} cycles per kv tile (matrix pipe: 512 per wave); "
                  f"whole loop {total} ticks")
            for nme, v in zip(SEG, per):
                print(f"    {v:7.0f}  {100 * v / tot:5.1f} %  {nme}")


if __name__ == "__main__":
    main()
