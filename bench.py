#!/usr/bin/env python
"""bench.py — regions/sec of the GAR region-captioning hot path (BASELINE.json metric) on N MI355X GPUs.

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of B synthetic regions per GPU: ViT over B*17 tiles, projector,
pool, embed-assemble + RoI replay, Llama prefill (S ~ 4.66k) and 64 greedy decode steps, then the RCCL gather of
the caption ids to rank 0. Inputs (pixel_values, mask values, input_ids) are resident in HBM before the timed
region starts. Weights: seeded synthetic GAR-1B (no checkpoint is reachable offline), bf16.

One JSON line on rank 0 (see DESIGN.md "Measurement"): value = N*K*B / max-over-ranks wall time; `roofline` = the
dominant kernel (bf16 tile GEMM: ViT + projector + prefill) timed live with HIP events on the launch stream over the
timed region; `cpu_baseline` = the fp32 CPU oracle on a bounded sample of the same workload on this box's cores.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("GAR_BENCH_BATCH", "64")),
                    help="regions per step per GPU (continuous batching of independent regions)")
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--max-num-tiles", type=int, default=16)
    ap.add_argument("--model", default="gar_1b")
    ap.add_argument("--pool", type=int, default=2, help="distinct pre-staged synthetic samples per rank")
    ap.add_argument("--preprocess", choices=["resident", "device"], default="resident",
                    help="resident (default, the bench contract): model inputs already in HBM. device: every step also "
                         "builds its B samples from host PIL images + masks (id matrix, bbox, prompt ids, H2D of the "
                         "raw uint8 image, resize/tile/normalise kernels of preprocess.hip) inside the timed region")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    return ap.parse_args()


def build_batches(cfg, proc, rank, world, B, pool, device):
    """`pool` distinct batches of B samples each, already on the GPU in bf16 (seeded per global region index)."""
    from gar_amd.eval_dataset import SingleRegionCaptionDataset
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    singles = []
    for j in range(pool * B if pool * B <= 4 else 4):          # host bicubic preprocessing is slow: 4 distinct images
        i = rank * 1000 + j
        singles.append(SingleRegionCaptionDataset(synthetic_image(i), synthetic_mask(i), proc,
                                                  data_dtype=torch.bfloat16, device="cpu")[0])
    batches = []
    for pidx in range(pool):
        sel = [singles[(pidx * B + k) % len(singles)] for k in range(B)]
        batches.append(dict(
            input_ids=torch.cat([s["input_ids"] for s in sel]).to(device),
            pixel_values=torch.cat([s["pixel_values"] for s in sel]).to(device),
            global_mask_values=torch.cat([s["global_mask_values"] for s in sel]).to(device),
            bboxes=[s["bboxes"][0] for s in sel],
            aspect_ratios=torch.cat([s["aspect_ratios"] for s in sel]).to(device)))
    return batches, singles[0]


def cpu_baseline(cfg, W, sample, new_tokens, threads):
    """fp32 CPU oracle ("port") on a bounded sample of ONE region of the same workload:
    2 of the 17 ViT tiles through all layers + projector, the full prefill, 4 decode steps; stage times are scaled to
    the full region (17 tiles, `new_tokens` tokens) — about 10-30 s of CPU work."""
    from oracle import gar_oracle as O
    if threads > 0:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    pv = sample["pixel_values"].float()
    mv = sample["global_mask_values"].float()
    T = pv.shape[0]
    nt = 2
    t0 = time.perf_counter()
    binary = O.decode_mask_values(mv[:nt], cfg.prompt_numbers)
    me = O.mask_patch_embed(binary, W["mask_patch_embedding.weight"])
    feats = O.get_image_features(pv[:nt], me, W, cfg, "sdpa")
    t_vit = (time.perf_counter() - t0) * (T / nt)
    t0 = time.perf_counter()
    full = feats[:1].repeat(T, 1, 1)                               # placeholder features: timing only
    emb = O.embed_and_scatter(sample["input_ids"], W[O.LM + "embed_tokens.weight"], full, cfg.mllm_config.image_token_id)
    emb = O.feature_replay(emb, sample["input_ids"], full, sample["aspect_ratios"], sample["bboxes"], cfg)
    t_asm = time.perf_counter() - t0
    tcfg = cfg.mllm_config.text_config
    cache = O.KVCache(tcfg.num_hidden_layers)
    t0 = time.perf_counter()
    h = O.llama_forward(emb, W, tcfg, cache, "sdpa")
    t_pre = time.perf_counter() - t0
    nd = 4
    t0 = time.perf_counter()
    head = O.lm_head_weight(W, tcfg)
    for _ in range(nd):
        nxt = torch.argmax(torch.nn.functional.linear(h[:, -1], head), -1)
        h = O.llama_forward(torch.nn.functional.embedding(nxt, W[O.LM + "embed_tokens.weight"]).unsqueeze(1), W, tcfg,
                            cache, "sdpa")
    t_dec = (time.perf_counter() - t0) * (new_tokens / nd)
    total = t_vit + t_asm + t_pre + t_dec
    return {"value": 1.0 / total, "unit": "regions/s", "cores": cores, "kind": "port",
            "sample": f"1 region: {nt}/{T} ViT tiles x all layers (x{T / nt:.1f}), full embed+RoI replay, full prefill "
                      f"S={emb.shape[1]}, {nd}/{new_tokens} decode steps (x{new_tokens / nd:.0f}); fp32 torch-CPU oracle",
            "seconds_per_region": total,
            "stage_seconds": {"vit+projector": t_vit, "assemble+replay": t_asm, "prefill": t_pre, "decode": t_dec}}


def main():
    args = parse()
    from gar_amd import GARConfig, dp, ops
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    from gar_amd.weights import synthetic_weights
    rank, local, world = dp.init_distributed()
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (launch with torch.distributed.run)"
    device = f"cuda:{local}"
    torch.cuda.set_device(local)
    cfg = getattr(GARConfig, args.model)()
    proc = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles)
    W = None
    if rank == 0:
        W = synthetic_weights(cfg, seed=0)
        model = GARModel(cfg, W, torch.bfloat16, device)
    else:
        model = GARModel.from_shapes(cfg, torch.bfloat16, device)
    model.broadcast_weights(src=0)                                   # RCCL broadcast over xGMI (no-op at N=1)
    batches, one = build_batches(cfg, proc, rank, world, args.batch, args.pool, device)
    B = args.batch
    S = batches[0]["input_ids"].shape[1]
    tiles = batches[0]["pixel_values"].shape[0] // B

    if args.preprocess == "device":
        from gar_amd.eval_dataset import SingleRegionCaptionDataset
        from gar_amd.synthetic import synthetic_image, synthetic_mask
        gproc = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles).use_gpu_preprocessing(device, torch.bfloat16)
        raw = [(synthetic_image(rank * 1000 + j), synthetic_mask(rank * 1000 + j)) for j in range(4)]

        def make_batch(i):
            sel = [SingleRegionCaptionDataset(*raw[(i * B + k) % len(raw)], gproc, data_dtype=torch.bfloat16,
                                              device=device)[0] for k in range(B)]
            return dict(input_ids=torch.cat([s["input_ids"] for s in sel]),
                        pixel_values=torch.cat([s["pixel_values"] for s in sel]),
                        global_mask_values=torch.cat([s["global_mask_values"] for s in sel]),
                        bboxes=[s["bboxes"][0] for s in sel],
                        aspect_ratios=torch.cat([s["aspect_ratios"] for s in sel]))
    else:
        def make_batch(i):
            return batches[i % len(batches)]

    pool = None
    if args.preprocess == "device":
        # pipeline: a helper thread builds the batch of step i+1 (host work + uploads + resize kernels, enqueued on the
        # same stream) while the main thread drives step i; exactly one batch is built per step inside the timed region
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(1)

        side = torch.cuda.Stream(device=device)

        def _build(i):
            torch.cuda.set_device(local)
            with torch.cuda.stream(side):            # uploads + resize kernels overlap the main stream's step
                b = make_batch(i)
                ev = torch.cuda.Event()
                ev.record(side)
            return b, ev
        pending = [pool.submit(_build, 0)]

    def step(i):
        if pool is not None:
            batch, ev = pending.pop().result()
            main = torch.cuda.current_stream()
            main.wait_event(ev)
            for t in batch.values():
                if torch.is_tensor(t):
                    t.record_stream(main)
            pending.append(pool.submit(_build, i + 1))
        else:
            batch = make_batch(i)
        out = model.generate(**batch, max_new_tokens=args.new_tokens, eos_token_id=None, validate=False)
        return dp.gather_captions(out.sequences, dst=0)

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    ops.KERNEL_TIMERS = []
    t0 = time.perf_counter()
    for i in range(args.steps):
        caps = step(args.warmup + i)
    torch.cuda.synchronize()
    dp.barrier()
    torch.cuda.synchronize()
    elapsed = dp.max_over_ranks(time.perf_counter() - t0, device)
    timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS, None

    if rank != 0:
        return
    # ---- roofline of the dominant kernel (bf16 tile GEMM), live HIP-event timing over the timed region ---------------
    agg = {}
    for kind, flops, nbytes, e0, e1 in timers:
        a = agg.setdefault(kind, [0.0, 0.0, 0.0, 0])
        a[0] += flops
        a[1] += nbytes
        a[2] += e0.elapsed_time(e1) * 1e-3
        a[3] += 1
    roof = None
    if "gemm_tile_bf16" in agg:
        fl, nb, sec, cnt = agg["gemm_tile_bf16"]
        roof = {"kernel": "gemm_bf16_pp_kernel (ViT qkv/proj/fc1/fc2, patch-embed, projector, Llama prefill qkv/o/gate-up/down)",
                "bound": "mfma", "achieved": fl / sec / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / sec / 1e12 / PEAK_BF16_TFLOPS, "traffic": None,
                "launches": cnt, "avg_launch_us": sec / cnt * 1e6, "flop_per_launch": fl / cnt,
                "algorithmic_bytes_per_launch": nb / cnt, "time_share_of_step": sec / elapsed}
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE and
        # WRITE_SIZE cannot share a pass), folded by tools/pmc_summary.py and committed under profiles/
        pmc = os.path.join(ROOT, "profiles", "r1_pmc_traffic.json")
        if os.path.exists(pmc):
            try:
                k = json.load(open(pmc))["kernels"]["gemm_bf16_pp_kernel"]
                roof["traffic"] = k["traffic_bytes_per_launch"]
                roof["traffic_source"] = "profiles/r1_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
            except Exception:
                pass
    value = world * args.steps * B / elapsed
    mname = {"gar_1b": "GAR-1B", "gar_8b": "GAR-8B"}.get(args.model, args.model)
    cfg_idx = {"gar_1b": "configs[1]", "gar_8b": "configs[3] shape, one GPU"}.get(args.model, "parity config")
    line = {"metric": f"regions/sec (1024^2 img, 1 mask, 64-tok caption) {mname}", "value": value, "unit": "regions/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": f"{mname} bf16, synthetic 1024x1024 images, 1 mask/region, 64-token greedy caption "
                                   f"(BASELINE.json {cfg_idx})",
                       "regions_per_step_per_gpu": B, "tiles_per_region": tiles, "prefill_len": S,
                       "new_tokens": args.new_tokens, "max_num_tiles": args.max_num_tiles,
                       "inputs": "resident in HBM" if args.preprocess == "resident" else
                                 "built per step from host images (device preprocessing inside the timed region)",
                       "weights": f"seeded synthetic {mname}", "parallelism": f"dp{world} (replica per GPU, RCCL weight "
                                                                              f"broadcast + caption gather)"},
            "roofline": roof}
    if "gemm_skinny_bf16" in agg:
        fl, nb, sec, cnt = agg["gemm_skinny_bf16"]
        line["roofline_other"] = {"gemm_skinny_bf16(prefill head only; decode runs inside the hipGraph)":
                                  {"bound": "hbm", "achieved": nb / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                                   "frac": nb / sec / 1e9 / PEAK_HBM_GBS, "launches": cnt}}
    # HBM-bound "replay pass" (pool -> embed + scatter -> RoI replay; SURVEY.md section 8d: ~90 MB per region)
    hb = [agg[k] for k in ("pool2x2", "embed_assemble", "roi_replay") if k in agg]
    if hb:
        nb, sec = sum(a[1] for a in hb), sum(a[2] for a in hb)
        line.setdefault("roofline_other", {})["pool2x2 + embed_assemble + roi_replay pass"] = {
            "bound": "hbm", "achieved": nb / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": nb / sec / 1e9 / PEAK_HBM_GBS, "bytes_per_region": nb / (args.steps * B),
            "launches": sum(a[3] for a in hb)}
    if "roi_replay" in agg:
        fl, nb, sec, cnt = agg["roi_replay"]
        line.setdefault("roofline_other", {})["roi_replay_batched_kernel alone (one launch per 16-region chunk)"] = {
            "bound": "hbm", "achieved": nb / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": nb / sec / 1e9 / PEAK_HBM_GBS, "bytes_per_launch": nb / cnt, "avg_launch_us": sec / cnt * 1e6,
            "launches": cnt}
    if not args.no_cpu_baseline and world == 1:          # reported on rank 0 at N=1 only
        try:
            line["cpu_baseline"] = cpu_baseline(cfg, W, one, args.new_tokens, args.cpu_threads)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU result
            line["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
