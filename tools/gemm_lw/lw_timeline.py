#!/usr/bin/env python
"""Timeline of the 4-wave tile GEMM (csrc/gemm_lw.hip) — diagnostic build `tools/build_variant.sh lwtl gemm_lw "-DLW_TIMELINE -DLW_MIN_K=256"`:
workgroup 0 sums, per wave, the shader-clock ticks between its stamps and writes them through `pos` (tokens_in = -777).
    GAR_HIP_LIB=grasp-any-region_amd/gar_amd/variants/libgar_hip_lwtl.so python tools/lw_timeline.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def main():
    hip.require_device(0)
    dev = "cuda:0"
    shapes = [("llm gate/up", 122668, 16384, 2048, hip.EPI_SWIGLU), ("llm down", 122668, 2048, 8192, hip.EPI_RES),
              ("llm o", 122668, 2048, 2048, hip.EPI_RES), ("vit fc2", 396675, 1024, 4096, hip.EPI_NONE),
              ("vit proj", 396675, 1024, 1024, hip.EPI_NONE)]
    for name, M, N, K, epi in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        No = N // 2 if epi == hip.EPI_SWIGLU else N
        out = torch.zeros(M, No, device=dev, dtype=torch.bfloat16)
        dbg = torch.zeros(64, dtype=torch.int32, device=dev)
        kw = dict(pos=dbg, tokens_in=-777)
        if epi == hip.EPI_RES:
            kw["residual"] = out
        for _ in range(2):
            ops.gemm(a, w, out, epi, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dbg.zero_()
        e0.record()
        ops.gemm(a, w, out, epi, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        per_block = (tiles + 255) // 256
        nk = K // 64
        d = (dbg.cpu().view(8, 8)[:4].to(torch.int64) & 0xffffffff).double()
        print(f"--- {name}: M={M} N={N} K={K}  {ms:.3f} ms = {2.0 * M * N * K / ms / 1e9:.0f} TFLOP/s; workgroup 0 ran {per_block} tiles x {nk} K tiles, "
              f"{float(d[0][7]):.0f} ticks -> {float(d[0][7]) / ms / 1e6:.2f} GHz if it spans the kernel")
        print("wave   k-step 0   vmcnt wait   barrier wait   k-step 1   cursor   | per K tile      tile top   epilogue | per output tile     main-loop share")
        for wv in range(4):
            x = d[wv]
            kt = per_block * nk
            per_k = [float(x[i]) / kt for i in range(5)]
            top, epi_t = float(x[5]) / per_block, float(x[6]) / per_block
            main = sum(float(x[i]) for i in range(5))
            print(f"  {wv}   {per_k[0]:8.0f}   {per_k[1]:10.0f}   {per_k[2]:12.0f}   {per_k[3]:8.0f}   {per_k[4]:6.0f}   = {sum(per_k):7.0f}      "
                  f"{top:8.0f}   {epi_t:8.0f}                        {main / float(x[7]):.3f}")


if __name__ == "__main__":
    main()
