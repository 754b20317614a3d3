#!/usr/bin/env python
"""Segment timeline of attn_bf16_v4_kernel (diagnostic build -DV4_TIMELINE, see tools/build_variant.sh):

    tools/build_variant.sh v4tl attention_v4 -DV4_TIMELINE
    GAR_HIP_LIB=grasp-any-region_amd/gar_amd/variants/libgar_hip_v4tl.so python tools/attn_v4_timeline.py

Workgroup 0 stamps s_memtime between the segments of every tile iteration of every item it walks and writes its per-wave sums
into rows 0..3 of O. One tile of one wave = 32 v_mfma_f32_32x32x16_bf16 = 1024 matrix-pipe cycles of its SIMD (one wave per SIMD).
s_memtime ticks at 100 MHz: ticks x (shader clock / 100 MHz) = cycles; the tool prints ticks and ns."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402

SEG = ["top: flags, exact prepare of q-block 0 (first / masked tiles)", "phase A: 16 MFMA | softmax qb0 | 24 fragment reads",
       "lazy check qb0, exact prepare of q-block 1", "phase B: 16 MFMA | softmax qb1 | tile DMA", "lazy check qb1, last-tile PV, cursors",
       "vmcnt wait (tile g + 2 landed)", "s_barrier"]


def main():
    hip.require_device(0)
    dev, dt = "cuda:0", torch.bfloat16
    for name, B, Hq, Hkv, hd, n, causal in (("vit", 272, 16, 16, 64, 1025, False), ("prefill", 16, 32, 8, 64, 4718, True)):
        npad = (n + 63) // 64 * 64
        Q = torch.randn(B, Hq, npad, hd, device=dev).to(dt) * 0.2
        K = torch.randn(B, Hkv, npad, hd, device=dev).to(dt)
        V = torch.randn(B, Hkv, npad, hd, device=dev).to(dt)
        O = torch.zeros(B * n, Hq * hd, device=dev, dtype=dt)
        pfx = 1 if (not causal and n % 64 == 1) else 0

        def run():
            ops.attention(Q, K, V, O, B, Hq, Hkv, hd, n, npad, n, npad, causal=causal, v_row_major=True, kv_prefix=pfx)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        fl = 4.0 * B * Hq * n * n * hd * (0.5 if causal else 1.0)
        raw = O[:4].contiguous().view(torch.int32)[:, :16].cpu().to(torch.int64) & 0xffffffff
        print(f"--- {name}: B={B} Hq={Hq} Hkv={Hkv} n={n} causal={causal}; instrumented launch {ms:.3f} ms ({fl / ms / 1e9:.0f} TFLOP/s)")
        for w in range(4):
            r = raw[w].tolist()
            ntl, total, items = r[6], r[7], r[8]
            if ntl == 0:
                print(f"wave {w}: nothing recorded {r}")
                continue
            per = [x / ntl for x in r[:6]]
            tot = sum(per)
            print(f"wave {w}: {items} items, {ntl} tile iterations; {tot:7.1f} ticks = {tot * 10:6.0f} ns per tile iteration; whole kernel {total} ticks")
            for nme, v in zip(SEG[1:], per):
                print(f"    {v:7.1f} ticks  {100 * v / tot:5.1f} %  {nme}")


if __name__ == "__main__":
    main()
