#!/usr/bin/env python
"""Micro-benchmark of the bf16 GEMM kernels on the benchmark's shapes (random data, HIP-event timing).
Large problems take the 256x256 ping-pong kernel (gemm_pp.hip), small ones the 128x128 kernel. The decomposition quoted in DESIGN.md
section 9 ("nostore" = main loop only, "L2store" = every tile stores into the first tile's L2-resident region) comes from
diagnostic builds of the library: tools/build_variant.sh nostore gemm_pp -DPP_NOSTORE (or l2store / -DPP_L2STORE), then
GAR_HIP_LIB=.../variants/libgar_hip_nostore.so python tools/bench_gemm.py."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import hip, ops  # noqa: E402

SHAPES = [("vit qkv", 139400, 3072, 1024, hip.EPI_BIAS), ("vit proj", 139400, 1024, 1024, hip.EPI_BIAS_SCALE_RES),
          ("vit fc1", 139400, 4096, 1024, hip.EPI_BIAS_GELU), ("vit fc2", 139400, 1024, 4096, hip.EPI_BIAS_SCALE_RES),
          ("llm qkv", 37744, 3072, 2048, hip.EPI_NONE), ("llm o", 37744, 2048, 2048, hip.EPI_RES),
          ("llm gate/up", 37744, 16384, 2048, hip.EPI_SWIGLU), ("llm down", 37744, 2048, 8192, hip.EPI_RES),
          ("square 8k", 8192, 8192, 8192, hip.EPI_NONE),
          ("proj none", 139400, 1024, 1024, hip.EPI_NONE), ("proj bias", 139400, 1024, 1024, hip.EPI_BIAS),
          ("qkv none", 139400, 3072, 1024, hip.EPI_NONE), ("llm o none", 37744, 2048, 2048, hip.EPI_NONE)]

# the planner's passes (gar_amd/planner.py, bench default): 387 image tiles x 1025 tokens, 26 sequences x 4718 tokens
PLAN_SHAPES = [("vit qkv rope", 396675, 3072, 1024, hip.EPI_QKV_ROPE), ("vit qkv", 396675, 3072, 1024, hip.EPI_BIAS), ("vit proj", 396675, 1024, 1024, hip.EPI_BIAS_SCALE_RES),
               ("vit fc1", 396675, 4096, 1024, hip.EPI_BIAS_GELU), ("vit fc2", 396675, 1024, 4096, hip.EPI_BIAS_SCALE_RES),
               ("llm qkv rope", 122668, 3072, 2048, hip.EPI_QKV_ROPE_LLM),
               ("llm qkv", 122668, 3072, 2048, hip.EPI_NONE), ("llm o", 122668, 2048, 2048, hip.EPI_RES),
               ("llm gate/up", 122668, 16384, 2048, hip.EPI_SWIGLU), ("llm down", 122668, 2048, 8192, hip.EPI_RES)]


def main():
    hip.require_device(0)
    dev = "cuda:0"
    reps = int(os.environ.get("REPS", "5"))
    # PAD_A / PAD_C: extra elements in the row pitch of the A operand / of the output (+ residual): row-strided views
    # (channel-camping experiment: power-of-two pitches put the 256 rows of a K tile on few HBM channels)
    pad_a, pad_c = int(os.environ.get("PAD_A", "0")), int(os.environ.get("PAD_C", "0"))
    tot_f = tot_t = 0.0
    shapes = PLAN_SHAPES if os.environ.get("SHAPESET") == "plan" else SHAPES
    for name, M, N, K, epi in shapes[:int(os.environ.get("SHAPES", "99"))]:
        a = torch.randn(M, K + pad_a, device=dev).to(torch.bfloat16)[:, :K]
        w = (torch.randn(N, K, device=dev) * K ** -0.5).to(torch.bfloat16)
        No = N // 2 if epi == hip.EPI_SWIGLU else N
        out = torch.empty(M, No + pad_c, device=dev, dtype=torch.bfloat16)[:, :No]
        kw = {}
        fold = os.environ.get("FOLD") == "1"          # folded norms: row_scale on the consumers, row_stats on the producers
        rs = torch.rand(M, device=dev) + 0.5 if fold and epi in (hip.EPI_NONE, hip.EPI_BIAS, hip.EPI_BIAS_GELU, hip.EPI_SWIGLU,
                                                                    hip.EPI_QKV_ROPE, hip.EPI_QKV_ROPE_LLM) else None
        if fold and epi in (hip.EPI_RES, hip.EPI_BIAS_SCALE_RES):
            kw["row_stats"] = torch.empty(M, (N + 63) // 64, 2, device=dev, dtype=torch.float32)
        if rs is not None and epi not in (hip.EPI_QKV_ROPE, hip.EPI_QKV_ROPE_LLM):
            kw["row_scale"] = rs
        if epi in (hip.EPI_BIAS, hip.EPI_BIAS_GELU, hip.EPI_BIAS_SCALE_RES, hip.EPI_QKV_ROPE):
            kw["bias"] = torch.randn(N, device=dev).to(torch.bfloat16)
        if epi in (hip.EPI_BIAS_SCALE_RES, hip.EPI_RES):
            kw["residual"] = out
        if epi == hip.EPI_BIAS_SCALE_RES:
            kw["gamma"] = torch.randn(N, device=dev).to(torch.bfloat16)
        if epi == hip.EPI_QKV_ROPE:         # the fused ViT qkv GEMM: 16 heads x 64, 1025 tokens per image tile (1 cls), v head-major
            H, hd, T = 16, 64, M // 1025
            Q_, K_, V_ = (torch.zeros(T, H, 1088, hd, device=dev, dtype=torch.bfloat16) for _ in range(3))
            ang = torch.randn(1024, hd // 2, device=dev)
            sin, cos = (f(ang).repeat_interleave(2, -1).contiguous() for f in (torch.sin, torch.cos))

            def call():
                assert ops.gemm_qkv_rope(a, w, kw["bias"], out, Q_, K_, sin, cos, H, hd, 1025, 1088, 1, 0.18, V=V_, row_scale=rs)
        elif epi == hip.EPI_QKV_ROPE_LLM:   # the fused Llama prefill qkv GEMM: 32 q / 8 kv heads x 64, 26 sequences of 4718 tokens
            Hq, Hkv, hd, S = 32, 8, 64, 4718
            B_ = M // S
            Q_ = torch.zeros(B_, Hq, 4736, hd, device=dev, dtype=torch.bfloat16)
            K_, V_ = (torch.zeros(B_, Hkv, 4864, hd, device=dev, dtype=torch.bfloat16) for _ in range(2))
            ang = torch.randn(4864, hd // 2, device=dev)
            cos, sin = torch.cos(ang).contiguous(), torch.sin(ang).contiguous()

            def call():
                assert ops.gemm_qkv_rope_llm(a, w, Q_, K_, V_, cos, sin, B_, S, 4736, Hq, Hkv, hd, 4864, 0, None, 0.18, row_scale=rs)
        else:
            def call():
                ops.gemm(a, w, out, epi, **kw)
        for _ in range(2):
            call()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            call()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / reps
        fl = 2.0 * M * N * K
        tot_f += fl
        tot_t += ms
        print(f"{name:12s} M={M:6d} N={N:5d} K={K:4d}  {ms:8.3f} ms  {fl / ms / 1e9:8.1f} TFLOP/s", flush=True)
    print(f"weighted: {tot_f / tot_t / 1e9:.1f} TFLOP/s  (library: {os.environ.get('GAR_HIP_LIB', 'product')}, PAD_A={pad_a} PAD_C={pad_c})")


if __name__ == "__main__":
    main()
