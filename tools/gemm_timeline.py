#!/usr/bin/env python
"""Per-phase timeline of the ping-pong GEMM main loop (diagnostic). Needs a library built with
`make EXTRA_gemm_pp=-DPP_TIMELINE`: block 0 accumulates, per wave, the shader-clock cycles spent working before and
waiting at each of the eight barriers of a K tile (L0 M0 L1 M1 L2 M2 L3 M3; L = ds_reads + DMA issue, M = 16 MFMAs)
and writes the sums through `pos` (tokens_in = -777). Prints cycles per K tile."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402


def main():
    hip.require_device(0)
    dev = "cuda:0"
    # store behaviour (real / none / L2-resident) is a property of the library build: see tools/build_variant.sh
    shapes = [("proj", 139400, 1024, 1024), ("qkv", 139400, 3072, 1024), ("llm down", 37744, 2048, 8192)]
    if os.environ.get("MORE_SHAPES"):   # does the slow start of an output tile follow the A panel being cold?
        shapes += [("proj, A (64 MB) resident in the Infinity Cache", 32768, 1024, 1024),
                   ("fc1: every A panel shared by 16 n-tiles", 139400, 4096, 1024),
                   ("fc2", 139400, 1024, 4096), ("llm o", 37744, 2048, 2048)]
    for name, M, N, K in shapes:
        a = torch.randn(M, K, device=dev).to(torch.bfloat16)
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        dbg = torch.zeros(8 * 16 + 1, dtype=torch.int32, device=dev)
        kw = dict(pos=dbg, tokens_in=-777)
        ops.gemm(a, w, out, hip.EPI_NONE, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dbg.zero_()
        e0.record()
        ops.gemm(a, w, out, hip.EPI_NONE, **kw)
        e1.record()
        torch.cuda.synchronize()
        wall_ms = e0.elapsed_time(e1)
        tiles = ((M + 255) // 256) * ((N + 255) // 256)
        per_block = (tiles + 255) // 256                    # output tiles block 0 ran
        ktiles = per_block * (K // 64)
        total_ticks = float(dbg[128].item() & 0xffffffff)
        dbg = dbg[:128]
        d = dbg.cpu().view(8, 16).double() / ktiles
        print(f"--- {name}: M={M} N={N} K={K}  block 0 ran {per_block} tiles = {ktiles} K tiles; cycles per K tile;  kernel "
              f"{wall_ms:.3f} ms wall, block 0 {total_ticks:.0f} ticks -> {total_ticks / wall_ms / 1e6:.2f} GHz if block 0 spans the kernel")
        print("wave  " + "".join(f"{p + ('' if i % 2 == 0 else 'w'):>7s}" for p in
                                  ("L0", "M0", "L1", "M1", "L2", "M2", "L3", "M3") for i in range(2)) + "    total")
        for wv in range(8):
            if os.environ.get("EPILOGUE"):       # -DPP_TIMELINE=5: epilogue phases, ticks per output tile (4 chunks)
                x = dbg.cpu().view(8, 16).double()[wv] / per_block
                print(f"  {wv}   un-stagger + aux loads + LDS write {float(x[0]):7.0f} | barrier {float(x[1]):7.0f} | LDS read + math + "
                      f"store issue {float(x[2]):7.0f} | barrier {float(x[3]):7.0f}   [ticks per output tile]")
            elif os.environ.get("TILEPOS"):        # -DPP_TIMELINE=4: K-tile duration by position in the output tile
                x = dbg.cpu().view(8, 16).double()[wv]
                nk = K // 64
                n_later = max(nk - 5, 0)             # K tiles 4 .. nk-2 land in the "later" bucket
                print(f"  {wv}   K tile 0 {float(x[8]) / per_block:7.0f}  1 {float(x[9]) / per_block:7.0f}  2 {float(x[10]) / per_block:7.0f}  "
                      f"3 {float(x[11]) / per_block:7.0f}  later (avg) {float(x[12]) / per_block / max(n_later, 1):7.0f}  "
                      f"un-stagger + epilogue {float(x[13]) / per_block:8.0f}   [ticks per output tile]")
            elif os.environ.get("LIGHT"):          # -DPP_TIMELINE=2: only phase 0 and the whole K tile are stamped
                print(f"  {wv}   L0 {float(d[wv][0]):6.0f}  L0w {float(d[wv][1]):6.0f}  K tile {float(d[wv][15]):6.0f}   "
                      f"[-DPP_TIMELINE=3, phase 3 after barrier 6: 16 MFMAs issued {float(d[wv][2]):5.0f}, next-tile address "
                      f"prep {float(d[wv][3]):5.0f}, vmcnt wait {float(d[wv][4]):5.0f}, wait at barrier 7 {float(d[wv][5]):5.0f}]")
            else:
                print(f"  {wv}   " + "".join(f"{x:7.0f}" for x in d[wv].tolist()) + f"  {float(d[wv].sum()):7.0f}")


if __name__ == "__main__":
    main()
