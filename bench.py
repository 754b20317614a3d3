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

HALF_DT = torch.bfloat16      # element type of the timed path; --data-type fp16 switches it (main)

PEAK_BF16_TFLOPS = 2500.0     # dense bf16 MFMA peak, /opt/skills/guides/MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--data-type", choices=["bf16", "fp16"], default="bf16",
                    help="element type of the timed path: bf16 (BASELINE.json's, the default) or fp16 (the twin library; a "
                         "secondary line, never the headline)")
    ap.add_argument("--batch", type=int, default=64,
                    help="regions per step per GPU (continuous batching of independent regions)")
    ap.add_argument("--prefill-chunk", type=int, default=0,
                    help="0 (default): image tiles per vision-tower pass and sequences per prefill pass chosen by "
                         "gar_amd/planner.py (whole rounds of the persistent tile GEMM); n > 0: both passes over chunks "
                         "of n regions")
    ap.add_argument("--overlap-decode", action="store_true",
                    help="run every step's decode loop on a second stream beside the NEXT step's prompt phase (GenerationPipeline, "
                         "two KV-state slots) instead of behind its own prompt phase: +0.7 ... 0.8 %% regions/s on one box — the "
                         "persistent tile GEMM owns the CUs, little co-schedules — at the price of per-kernel timings (roofline "
                         "fields) that include the other stream's kernels; off by default")
    ap.add_argument("--no-graph", action="store_true",
                    help="decode loop with eager launches instead of hipGraph replays (needed under rocprofv3 --pmc)")
    ap.add_argument("--new-tokens", type=int, default=64)
    ap.add_argument("--max-num-tiles", type=int, default=16)
    ap.add_argument("--workload", choices=["single", "multi_region", "video"], default="single",
                    help="single: BASELINE.json configs[1] (1 mask / region, the headline metric); multi_region: configs[2] "
                         "(4 masks per image, relationship prompt, demo/gar_relationship.py path); video: configs[4] "
                         "shape on one GPU (8-frame 1024^2 clip, per-frame mask, GAR-8B video replay)")
    ap.add_argument("--model", default=None, help="gar_1b | gar_8b (default: gar_8b for --workload video, else gar_1b)")
    ap.add_argument("--pool", type=int, default=2, help="pre-staged batches per rank (each of --batch distinct regions)")
    ap.add_argument("--distinct-samples", type=int, default=0,
                    help="A/B: cap on the number of different synthetic samples per rank (0 = every region of every batch "
                         "is its own image + mask, the default)")
    ap.add_argument("--preprocess", choices=["resident", "device"], default="resident",
                    help="resident (default, the bench contract): model inputs already in HBM. device: every step also "
                         "builds its B samples from host PIL images + masks (id matrix, bbox, prompt ids, H2D of the "
                         "raw uint8 image, resize/tile/normalise kernels of preprocess.hip) inside the timed region")
    ap.add_argument("--no-patch-gather", action="store_true",
                    help="A/B: patch-embed as gar_patch_im2col + GEMM instead of gar_patch_embed (patches DMA'd from the "
                         "image tiles into LDS)")
    ap.add_argument("--vit-v-transpose", action="store_true",
                    help="A/B: ViT v through gar_vit_v_transpose + Vt attention instead of the row-major form")
    ap.add_argument("--dry-run", action="store_true",
                    help="print the launch plan of --gpus N as one JSON object (command, per-rank device / regions / host threads / GPU "
                         "NUMA node, collectives) and exit: nothing is started, no GPU is touched")
    ap.add_argument("--runtime", default=None,
                    help="module:attr of a runtime object replacing GpuRuntime (tests/bench_stub.py: a CPU stub under gloo, so that "
                         "the N > 1 control flow and the self-launch run on a box without GPUs); never set for a measurement")
    ap.add_argument("--decode-probe-steps", type=int, default=2,
                    help="eager decode steps timed kernel by kernel AFTER the timed region (rank 0) for the roofline_other entries of "
                         "the kernels that run inside the hipGraph during it (decode attention, decode GEMVs); 0 = off")
    ap.add_argument("--vit-chunk-rows", type=int, default=0, help="A/B: cap on the token rows of one vision-tower pass (GARModel.VIT_CHUNK_ROWS)")
    ap.add_argument("--prefill-chunk-rows", type=int, default=0, help="A/B: cap on the rows of one prefill pass (GARModel.PREFILL_CHUNK_ROWS)")
    ap.add_argument("--no-prune-last-layer", action="store_true",
                    help="A/B: the last Llama prefill layer over all S rows (the reference's computation) instead of its last row only")
    ap.add_argument("--eos-mix", action="store_true",
                    help="secondary line: the reference's REAL calling pattern — generate until EOS under a token cap "
                         "(demo/gar_with_mask.py:112-122) — on a queue of 3 x --batch regions whose captions have MIXED lengths "
                         "(synthetic EOS ids picked from the regions' own token streams): static batches of --batch regions per "
                         "generate() against the continuous batcher (gar_amd/continuous.py: rows retire at EOS, queued regions are "
                         "admitted into the running decode loop)")
    ap.add_argument("--eos-mix-tokens", type=int, default=256, help="--eos-mix: max_new_tokens (the reference's callers use 1024)")
    ap.add_argument("--eos-mix-batches", type=int, default=3, help="--eos-mix: queue length in units of --batch regions")
    ap.add_argument("--eos-mix-admit-min", type=int, default=0,
                    help="--eos-mix: free rows the continuous batcher waits for before it runs a prompt phase (0 = its default, slots / 4)")
    ap.add_argument("--eos-mix-poll", type=int, default=8, help="--eos-mix: decode steps between two polls of the finished latches")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    a = ap.parse_args(argv)
    if a.model is None:
        a.model = "gar_8b" if a.workload == "video" else "gar_1b"
    if a.workload == "video" and a.max_num_tiles == 16:
        a.max_num_tiles = 8                      # GAR-8B's tile budget (configs/gar_8b.py:101); frames are one tile each
    return a


def build_sample(workload, proc, i, device="cpu"):
    """one synthetic sample of the workload, bf16 (seeded by the global region index i); with a device processor
    (GARProcessor.use_gpu_preprocessing) the tiles are resized / normalised on the GPU from the raw uint8 image."""
    from gar_amd.eval_dataset import MultiRegionDataset, SingleRegionCaptionDataset, VideoRegionCaptionDataset
    from gar_amd.synthetic import RELATIONSHIP_QUESTION, synthetic_disjoint_masks, synthetic_image, synthetic_mask
    if workload == "multi_region":       # 4 disjoint masks, prompt ids 0-3, fixed relationship question
        return MultiRegionDataset(synthetic_image(i), synthetic_disjoint_masks(i, 4), RELATIONSHIP_QUESTION, proc,
                                  data_dtype=HALF_DT, device=device)[0]
    if workload == "video":              # 8 frames of 1024^2, one mask per frame, one tile + one crop token per frame
        frames = [synthetic_image(8 * i + f) for f in range(8)]
        masks = [synthetic_mask(8 * i + f) for f in range(8)]
        return VideoRegionCaptionDataset(frames, masks, proc, data_dtype=HALF_DT, device=device)[0]
    return SingleRegionCaptionDataset(synthetic_image(i), synthetic_mask(i), proc, data_dtype=HALF_DT,
                                      device=device)[0]


def region_index(rank: int, world: int, j: int) -> int:
    """the global index of the j-th region a rank serves: regions are dealt round-robin, i % world == rank"""
    return rank + world * j


def launch_plan(args) -> dict:
    """`--dry-run`: what `--gpus N` WOULD start on this box — the launch command, and per rank its device, the regions it serves,
    its share of the host cores and the NUMA node of its GPU where the sysfs says — without touching a GPU."""
    n = args.gpus
    cores = os.cpu_count() or n
    ranks = []
    for r in range(n):
        numa = None
        try:        # /sys/class/drm/cardN/device/numa_node of the r-th render device, when present
            cards = sorted(d for d in os.listdir("/sys/class/drm") if d.startswith("card") and d[4:].isdigit())
            if r < len(cards):
                numa = int(open(f"/sys/class/drm/{cards[r]}/device/numa_node").read().strip())
        except (OSError, ValueError):
            pass
        ranks.append({"rank": r, "local_rank": r, "device": f"cuda:{r}", "torch_threads": max(1, cores // n),
                      "gpu_numa_node": numa,
                      "regions": f"i % {n} == {r}: " + ", ".join(str(region_index(r, n, j)) for j in range(3)) + ", ...",
                      "regions_per_step": args.batch})
    return {"dry_run": True, "n_gpus": n, "backend": "nccl (RCCL over xGMI)",
            "launch": f"{sys.executable} -m torch.distributed.run --nnodes=1 --nproc-per-node={n} --master-addr 127.0.0.1 "
                      f"--master-port <free port> {os.path.abspath(__file__)} --gpus {n} --steps {args.steps} --warmup {args.warmup}",
            "env": {"HSA_ENABLE_IPC_MODE_LEGACY": "0"},
            "collectives": {"weight_broadcast": "one in-place broadcast per dtype arena from rank 0 (2 for GAR-1B: bf16 + f32 tables)",
                            "per_step": f"gather of [{args.batch}, {args.new_tokens}] int64 caption ids to rank 0 -> [{n * args.batch}, {args.new_tokens}]",
                            "timed_region": "barrier + synchronize on both sides, max over ranks"},
            "ranks": ranks}


def build_batches(cfg, proc, rank, world, B, pool, device, workload="single", distinct=0):
    """`pool` batches of B samples each, resident on the GPU in bf16 before the timed region. Every region of every
    batch is its own synthetic image + mask (seeded by rank, batch and slot), built with the DEVICE preprocessor
    (csrc/preprocess.hip: bit-exact with the host processor, ~10 ms of host work per region instead of ~450 ms of CPU
    bicubic). ``distinct`` > 0 caps the number of different samples (they are then repeated; A/B only)."""
    n = pool * B if distinct <= 0 else min(pool * B, distinct)
    # region i of the job runs on rank i % world: rank r builds regions r, r + world, r + 2 world, ...
    singles = [build_sample(workload, proc, region_index(rank, world, j), device) for j in range(n)]
    batches = []
    for pidx in range(pool):
        sel = [singles[(pidx * B + k) % len(singles)] for k in range(B)]
        batches.append(dict(
            input_ids=torch.cat([s["input_ids"] for s in sel]).to(device),
            pixel_values=torch.cat([s["pixel_values"] for s in sel]).to(device),
            global_mask_values=torch.cat([s["global_mask_values"] for s in sel]).to(device),
            bboxes=[s["bboxes"][0] for s in sel],
            aspect_ratios=torch.cat([s["aspect_ratios"] for s in sel]).to(device)))
        if workload == "video":
            batches[-1].update(feature_replay_video=True, video_frame_tokens=sel[0]["video_frame_tokens"])
    one = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in singles[0].items()}
    return batches, one, n


def cpu_baseline(cfg, W, sample, new_tokens, threads):
    """fp32 CPU oracle ("port") on ONE region of the same workload, on this box's cores: ALL of its ViT tiles through all
    layers + projector, embed + RoI replay, the full prefill, and CPU_DECODE_STEPS of the `new_tokens` decode steps (the
    only extrapolated stage: a decode step's cost grows by one KV row per step, ~0.02 % here). The decode leg is timed at
    several intra-op thread counts first (one-row GEMVs do not want every SMT thread) and the best one is used and reported
    next to the maximum."""
    from oracle import gar_oracle as O
    max_threads = torch.get_num_threads()
    if threads > 0:
        torch.set_num_threads(threads)
    cores = torch.get_num_threads()
    pv = sample["pixel_values"].float()
    mv = sample["global_mask_values"].float()
    T = pv.shape[0]
    t0 = time.perf_counter()
    binary = O.decode_mask_values(mv, cfg.prompt_numbers)
    me = O.mask_patch_embed(binary, W["mask_patch_embedding.weight"])
    feats = O.get_image_features(pv, me, W, cfg, "sdpa")
    t_vit = time.perf_counter() - t0
    t0 = time.perf_counter()
    emb = O.embed_and_scatter(sample["input_ids"], W[O.LM + "embed_tokens.weight"], feats, cfg.mllm_config.image_token_id)
    emb = O.feature_replay(emb, sample["input_ids"], feats, sample["aspect_ratios"], sample["bboxes"], cfg)
    t_asm = time.perf_counter() - t0
    tcfg = cfg.mllm_config.text_config
    cache = O.KVCache(tcfg.num_hidden_layers)
    t0 = time.perf_counter()
    h = O.llama_forward(emb, W, tcfg, cache, "sdpa")
    t_pre = time.perf_counter() - t0
    head = O.lm_head_weight(W, tcfg)
    E = W[O.LM + "embed_tokens.weight"]

    def decode_steps(n):
        nonlocal h
        t = time.perf_counter()
        for _ in range(n):
            nxt = torch.argmax(torch.nn.functional.linear(h[:, -1], head), -1)
            h = O.llama_forward(torch.nn.functional.embedding(nxt, E).unsqueeze(1), W, tcfg, cache, "sdpa")
        return (time.perf_counter() - t) / n
    # thread-count probe for the decode leg (2 steps each), then the remaining steps at the best count
    cands = sorted({c for c in (cores, max(1, cores // 2), max(1, cores // 4), min(cores, 32), min(cores, 16)) if c >= 1},
                   reverse=True)
    probe = {}
    for c in cands:
        torch.set_num_threads(c)
        probe[c] = decode_steps(2)
    best = min(probe, key=probe.get)
    torch.set_num_threads(best)
    rest = max(0, CPU_DECODE_STEPS - 2 * len(cands))
    per_step = decode_steps(rest) if rest else probe[best]
    torch.set_num_threads(cores)
    nd = 2 * len(cands) + rest
    t_dec = per_step * new_tokens
    t_dec_max = probe[cores] * new_tokens
    total = t_vit + t_asm + t_pre + t_dec
    return {"value": 1.0 / total, "unit": "regions/s", "cores": cores, "kind": "port",
            "sample": f"1 region un-extrapolated except decode: {T}/{T} ViT tiles x all layers + projector, embed + RoI "
                      f"replay, full prefill S={emb.shape[1]}, {nd}/{new_tokens} decode steps timed (x{new_tokens / nd:.1f}); "
                      f"fp32 torch-CPU oracle",
            "extrapolated": "decode only",
            "extrapolation": {"vit+projector": 1.0, "assemble+replay": 1.0, "prefill": 1.0, "decode": new_tokens / nd},
            "decode_threads": {"best": best, "seconds_per_step": {str(k): round(v, 4) for k, v in probe.items()},
                               "value_at_max_threads": 1.0 / (t_vit + t_asm + t_pre + t_dec_max)},
            "note": "`cores` = torch intra-op threads of the GEMM-shaped stages (all hardware threads of the box); the "
                    "decode leg runs at `decode_threads.best` threads (probed; the figure at the maximum is next to it). "
                    "Its cost is mostly cache bookkeeping (torch.cat KV growth, a materialised K/V copy per GQA group) — "
                    "what the reference's eager CPU path does too (transformers DynamicCache.update, repeat_kv). A "
                    "reported baseline, not a tuned CPU implementation",
            "seconds_per_region": total,
            "stage_seconds": {"vit+projector": t_vit, "assemble+replay": t_asm, "prefill": t_pre, "decode": t_dec}}


CPU_DECODE_STEPS = 16


def parity_pins():
    """which pieces of the oracle are pinned by fixtures present in tests/golden/ (VERDICT r4 #7): the day a box has timm / torchvision
    and tools/capture_ext_goldens.py has been run, the record changes without anyone reading test logs"""
    g = os.path.join(ROOT, "tests", "golden")
    have = lambda *names: all(os.path.exists(os.path.join(g, n)) for n in names)
    pins = {"llama": have("llama_tiny.npz", "llama_tiny_padded.npz", "llama_inv_freq.npz"), "projector": have("projector_tiny.npz"),
            "helpers": have("ref_helpers.json", "rle_samples.json"), "torch_ops": have("torch_ops.npz"),
            "ext_timm_eva": have("ext_timm_eva.npz"), "ext_tv_roi_align": have("ext_tv_roi_align.npz")}
    pins["unpinned"] = [k for k, v in pins.items() if not v]
    pins["note"] = ("fixtures generated by the reference's own code / its pinned dependencies (tools/make_goldens.py, "
                    "tools/capture_ext_goldens.py); ext_* need timm==1.0.19 / torchvision, absent from this image: the ViT forward and "
                    "roi_align restatements of oracle/gar_oracle.py are parity-unpinned until they exist")
    return pins


def pick_synthetic_eos(streams, max_new, target_mean_frac=0.4, max_ids=12):
    """EOS ids for random-init weights: token ids out of the regions' own free-running streams, most widely shared first, until
    the captions they cut average ``target_mean_frac * max_new`` tokens — mixed lengths, from a few tokens to the cap."""
    first = {}
    for si, s in enumerate(streams):
        for j, t in enumerate(s):
            first.setdefault(t, {}).setdefault(si, j + 1)
    order = sorted(first, key=lambda t: (-len(first[t]), t))
    lens = [max_new] * len(streams)
    eos = []
    for t in order:
        if len(eos) >= max_ids or sum(lens) / len(lens) <= target_mean_frac * max_new:
            break
        new = [min(lens[si], first[t].get(si, max_new)) for si in range(len(streams))]
        if min(new) < 2 and sum(1 for n in new if n < 2) > len(streams) // 8:
            continue                    # would end many captions at their first token
        eos.append(t)
        lens = new
    return eos, lens


def eos_mix(args, model, make_batch, B, S, tiles, rt):
    """bench.py --eos-mix: see the flag's help. Both legs run the same kernels on the same regions; a region's
    caption is cut at its own EOS. Warm-up = one full pass of each leg (graph capture, workspaces)."""
    from gar_amd.continuous import ContinuousBatcher
    NM, NB = args.eos_mix_tokens, args.eos_mix_batches
    batches = [make_batch(i) for i in range(NB)]
    streams = model.generate(**batches[0], max_new_tokens=NM, eos_token_id=None, validate=False).sequences.cpu().tolist()
    eos, cal_lens = pick_synthetic_eos(streams, NM)

    def split(batch):
        out = []
        for b in range(B):
            out.append(dict(input_ids=batch["input_ids"][b:b + 1], pixel_values=batch["pixel_values"][b * tiles:(b + 1) * tiles],
                            global_mask_values=batch["global_mask_values"][b * tiles:(b + 1) * tiles],
                            bboxes=[batch["bboxes"][b]], aspect_ratios=batch["aspect_ratios"][b:b + 1]))
        return out
    regions = [r for bt in batches for r in split(bt)]

    def static_leg():
        caps, steps = [], 0
        for bt in batches:
            rows = model.generate(**bt, max_new_tokens=NM, eos_token_id=eos, validate=False, sync_every=8).sequences.cpu().tolist()
            steps += len(rows[0]) - 1
            for row in rows:
                caps.append(row[:next((j + 1 for j, t in enumerate(row) if t in eos), len(row))])
        return caps, steps

    def continuous_leg():
        cb = ContinuousBatcher(model, slots=B, max_new_tokens=NM, eos_token_id=eos, poll_every=args.eos_mix_poll,
                               admit_min=args.eos_mix_admit_min or None, validate=False)
        tickets = [cb.submit(r) for r in regions]
        res = cb.flush()
        return [res[t] for t in tickets], cb.stats

    def timed(fn):
        rt.sync()
        t0 = time.perf_counter()
        out = fn()
        rt.sync()
        return out, time.perf_counter() - t0
    static_leg()
    continuous_leg()
    (caps_s, steps_s), t_s = timed(static_leg)
    (caps_c, st), t_c = timed(continuous_leg)
    lens = [len(c) for c in caps_s]
    same = sum(1 for a, b in zip(caps_s, caps_c) if a == b)
    n = len(regions)
    need = sum(x - 1 for x in lens)
    return {"metric": f"regions/sec until EOS (cap {NM} tokens, mixed caption lengths) {args.model}", "unit": "regions/s",
            "value": n / t_c, "higher_is_better": True, "n_gpus": 1, "dtype": args.data_type, "data": "synthetic",
            "config": {"workload": f"queue of {n} regions (1024^2, 1 mask), {B} decode rows, EOS on, max_new_tokens {NM}", "S": S,
                       "synthetic_eos_ids": len(eos)},
            "continuous": {"regions_per_s": n / t_c, "seconds": t_c, "decode_steps": st["decode_steps"],
                           "prompt_passes": st["prompt_passes"], "row_occupancy": st["live_row_steps"] / max(1, st["row_steps"]),
                           "rebases": st["rebases"]},
            "static_batches": {"regions_per_s": n / t_s, "seconds": t_s, "decode_steps": steps_s},
            "speedup_vs_static": t_s / t_c,
            "caption_lengths": {"min": min(lens), "mean": sum(lens) / n, "max": max(lens),
                                "decode_tokens_needed": need, "lower_bound_steps": -(-need // B)},
            "captions_identical_to_static": f"{same}/{n}",
            "note": "static = generate() per batch of B regions until every row hit EOS (every row pays for the longest caption); "
                    "continuous = ContinuousBatcher. bf16: the two legs run different batch compositions per step, captions can "
                    "differ at near-tied steps (INTEGRATION.md)"}


class GpuRuntime:
    """What main() needs from the device side. The product path is this class; tests/test_bench_gloo.py substitutes a CPU
    stub (gloo, world 2) so that the N > 1 control flow — rank-0 weight build + broadcast, per-rank batches, the barrier /
    synchronize bracket, max-over-ranks, the caption gather and the rank-0 JSON line — is executed without a GPU."""
    backend = None                      # dp.init_distributed picks "nccl" (= RCCL) on a GPU box

    def device_of(self, local):
        torch.cuda.set_device(local)
        return f"cuda:{local}"

    def sync(self):
        torch.cuda.synchronize()

    def peak_mem_gib(self, device):
        return round(torch.cuda.max_memory_allocated(device) / 2 ** 30, 1)

    def build_model(self, args, cfg, rank, device):
        """rank 0 synthesises the weights; the other ranks allocate shapes only and receive them by broadcast"""
        from gar_amd.modeling_gar import GARModel
        from gar_amd.weights import synthetic_weights
        W = None
        if rank == 0:
            W = synthetic_weights(cfg, seed=0)
            model = GARModel(cfg, W, HALF_DT, device, prefill_chunk=args.prefill_chunk or None)
        else:
            model = GARModel.from_shapes(cfg, HALF_DT, device)
            model.prefill_chunk = args.prefill_chunk or None
        if args.no_patch_gather:
            model.w_patch_gather = None
        if args.vit_v_transpose:
            model.VIT_V_ROW_MAJOR = False
        if args.no_prune_last_layer:
            model.PRUNE_LAST_PREFILL_LAYER = False
        if args.vit_chunk_rows:
            model.VIT_CHUNK_ROWS = args.vit_chunk_rows
        if args.prefill_chunk_rows:
            model.PREFILL_CHUNK_ROWS = args.prefill_chunk_rows
        return model, W

    def build_batches(self, args, cfg, rank, world, device):
        from gar_amd.processing import GARProcessor
        dproc = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles).use_gpu_preprocessing(device, HALF_DT)
        return build_batches(cfg, dproc, rank, world, args.batch, args.pool, device, args.workload, args.distinct_samples)


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


_RENDEZVOUS_MARKERS = ("address already in use", "eaddrinuse", "errno 98", "rendezvousconnectionerror", "rendezvoustimeouterror",
                       "failed to bind", "the server socket has failed to listen", "connection refused")


def _is_rendezvous_failure(stderr_text: str) -> bool:
    low = stderr_text.lower()
    return any(m in low for m in _RENDEZVOUS_MARKERS)


def _run_tee_stderr(cmd, env):
    """run cmd; its stdout goes straight through, its stderr is passed through line by line AND its last 200 lines are returned"""
    import collections
    import subprocess
    import threading
    tail = collections.deque(maxlen=200)
    p = subprocess.Popen(cmd, env=env, stderr=subprocess.PIPE, text=True, errors="replace")

    def pump():
        for line in p.stderr:
            tail.append(line)
            sys.stderr.write(line)
            sys.stderr.flush()
    th = threading.Thread(target=pump, daemon=True)
    th.start()
    rc = p.wait()
    th.join(timeout=5.0)
    return rc, "".join(tail)


def self_launch(args, argv):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks ourselves — the same command the driver
    documents (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py
    ...`), one process per GPU — and pass their output and exit code through. The scaling line must not depend on how the caller
    spells the launch."""
    argv = list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC only on this host driver (RCCL needs it)
    rc = 1
    for attempt in range(3):
        # (a free port can be taken between this probe and the launcher's bind: ONLY a launch whose stderr shows a rendezvous /
        # address-in-use failure within seconds — before any work — is tried again on another port; every other failure, fast or
        # slow (bad flag, missing .so, out of memory at init), is passed through at once: ADVICE r4)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr",
               "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__), *argv]
        print(f"[bench] --gpus {args.gpus} without WORLD_SIZE: launching {' '.join(cmd)}", file=sys.stderr, flush=True)
        t0 = time.time()
        rc, err_tail = _run_tee_stderr(cmd, env)
        if rc == 0 or time.time() - t0 > 30.0 or not _is_rendezvous_failure(err_tail):
            break
    return rc


def load_runtime(spec):
    import importlib
    mod, _, attr = spec.partition(":")
    obj = getattr(importlib.import_module(mod), attr)
    return obj() if isinstance(obj, type) else obj


def main(argv=None, runtime=None):
    global HALF_DT
    args = parse(argv)
    HALF_DT = torch.float16 if args.data_type == "fp16" else torch.bfloat16
    if args.dry_run:
        print(json.dumps(launch_plan(args)), flush=True)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and runtime is None:
        rc = self_launch(args, argv)
        if rc:
            raise SystemExit(rc)
        return
    from gar_amd import GARConfig, dp, ops
    from gar_amd.processing import GARProcessor
    rt = runtime or (load_runtime(args.runtime) if args.runtime else GpuRuntime())
    rank, local, world = dp.init_distributed(rt.backend)
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's --nproc-per-node has to equal --gpus "
                         f"(a plain `python bench.py --gpus N` launches its own ranks)")
    device = rt.device_of(local)
    if world > 1:       # N ranks share the box's cores (weight synthesis on rank 0, sample building, tokenisation)
        torch.set_num_threads(max(1, (os.cpu_count() or world) // world))
    cfg = getattr(GARConfig, args.model)()
    proc = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles)
    model, W = rt.build_model(args, cfg, rank, device)
    rt.sync()
    dp.barrier()
    tb = time.perf_counter()
    bcast_n = model.broadcast_weights(src=0)                         # RCCL broadcast over xGMI (no-op at N=1)
    rt.sync()
    bcast_s = dp.max_over_ranks(time.perf_counter() - tb, device)
    bcast_bytes = model.weight_arena_bytes() if hasattr(model, "weight_arena_bytes") else None
    if args.workload != "single" and args.preprocess == "device":
        raise SystemExit("--preprocess device is wired for --workload single")
    batches, one, n_distinct = rt.build_batches(args, cfg, rank, world, device)
    B = args.batch
    S = batches[0]["input_ids"].shape[1]
    tiles = batches[0]["pixel_values"].shape[0] // B

    if args.preprocess == "device":
        from gar_amd.eval_dataset import SingleRegionCaptionDataset
        from gar_amd.synthetic import synthetic_image, synthetic_mask
        gproc = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles).use_gpu_preprocessing(device, HALF_DT)
        raw = [(synthetic_image(region_index(rank, world, j)), synthetic_mask(region_index(rank, world, j))) for j in range(4)]

        def make_batch(i):
            sel = [SingleRegionCaptionDataset(*raw[(i * B + k) % len(raw)], gproc, data_dtype=HALF_DT,
                                              device=device)[0] for k in range(B)]
            return dict(input_ids=torch.cat([s["input_ids"] for s in sel]),
                        pixel_values=torch.cat([s["pixel_values"] for s in sel]),
                        global_mask_values=torch.cat([s["global_mask_values"] for s in sel]),
                        bboxes=[s["bboxes"][0] for s in sel],
                        aspect_ratios=torch.cat([s["aspect_ratios"] for s in sel]))
    else:
        def make_batch(i):
            return batches[i % len(batches)]

    if args.eos_mix:
        if rank == 0:
            print(json.dumps(eos_mix(args, model, make_batch, B, S, tiles, rt)), flush=True)
        return

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
        if pipe is not None:            # this batch's prompt phase; the previous batch's decode loop runs beside it
            outs = pipe.submit(batch)
        else:
            outs = [model.generate(**batch, **gen_kw)]
        caps = None
        for out in outs:
            caps = collect(out)
        return caps

    def collect(out):
        if getattr(out, "done", None) is not None:      # produced on the pipeline's decode stream, consumed on this one
            torch.cuda.current_stream().wait_event(out.done)
            out.sequences.record_stream(torch.cuda.current_stream())
        input_flags.append(out.input_flags)             # device-side input checks: read after the timed region
        return dp.gather_captions(out.sequences, dst=0)

    def drain():
        caps = None
        if pipe is not None:
            for out in pipe.flush():
                caps = collect(out)
        return caps

    gen_kw = dict(max_new_tokens=args.new_tokens, eos_token_id=None, validate=False, use_graph=not args.no_graph)
    pipe = None
    if args.overlap_decode and hasattr(model, "generate_begin"):
        # the decode loop of step i on a second stream beside the prompt phase of step i + 1 (GenerationPipeline). Both KV-state
        # slots are allocated and their decode graphs captured here, before the warm-up steps
        from gar_amd.modeling_gar import GenerationPipeline
        for slot in (0, 1):
            model.generate(**make_batch(0), **gen_kw, state_slot=slot)
        pipe = GenerationPipeline(model, **gen_kw)
    input_flags = []
    for i in range(args.warmup):
        step(i)
    drain()
    rt.sync()
    dp.barrier()
    rt.sync()
    ops.KERNEL_TIMERS = []
    t0 = time.perf_counter()
    caps = None
    for i in range(args.steps):
        caps = step(args.warmup + i) or caps
    caps = drain() or caps              # the last step's decode loop: inside the timed region, like every other step's
    rt.sync()
    dp.barrier()
    rt.sync()
    my_elapsed = time.perf_counter() - t0
    elapsed = dp.max_over_ranks(my_elapsed, device)
    per_rank = dp.all_gather_floats(my_elapsed, device)             # every rank's own clock around the same K steps
    gathered_shape = None
    if rank == 0:       # every rank's [B, new_tokens] ids arrived on rank 0 in the last step
        assert caps is not None and len(caps) == world and all(tuple(c.shape) == (args.batch, args.new_tokens) for c in caps)
        gathered_shape = list(torch.cat([c.cpu() for c in caps]).shape)        # [world * B, new_tokens] on rank 0
    timers, ops.KERNEL_TIMERS = ops.KERNEL_TIMERS, None
    bad_inputs = 0
    for f in input_flags:
        bad_inputs |= int(f.item())
    if bad_inputs:
        from gar_amd.modeling_gar import describe_input_flags
        raise SystemExit(f"bench inputs failed the device-side checks: {describe_input_flags(bad_inputs)}")

    if rank != 0:
        return
    # ---- kernels that run inside the hipGraph during the timed region (decode attention, decode GEMVs, split-K reduce):
    # a few EAGER decode steps of the same batch after it, every launch bracketed by HIP events (rank 0)
    probe = []
    if args.decode_probe_steps > 0 and args.new_tokens > 1 and hasattr(model, "generate_finish"):
        ops.KERNEL_TIMERS = []
        model.generate(**make_batch(0), **{**gen_kw, "max_new_tokens": args.decode_probe_steps + 1, "use_graph": False})
        rt.sync()
        probe, ops.KERNEL_TIMERS = [t for t in ops.KERNEL_TIMERS if t[0].startswith("decode:")], None
    # ---- the decode loop by itself (what the reference's callers wait for per token at batch 1): one more generate after the
    # timed region, host clock around generate_finish() only — the hipGraph replays of new_tokens - 1 steps
    decode_loop = None
    if args.new_tokens > 1 and hasattr(model, "generate_finish"):
        pend = model.generate_begin(**make_batch(0), **gen_kw)
        rt.sync()
        td = time.perf_counter()
        model.generate_finish(pend)
        rt.sync()
        td = time.perf_counter() - td
        tcfg = cfg.mllm_config.text_config
        # the weights ONE decode step streams: a layer's qkv / o / gate-up / down (the folded forms where the model holds them;
        # keep_plain_weights=True holds both and must not count twice: ADVICE r4) + lm_head
        def _streamed(ly):
            ks = [("qkv_f" if "qkv_f" in ly else "qkv"), "o", ("gu_f" if "gu_f" in ly else "gu"), "down"]
            return sum(ly[k].numel() * ly[k].element_size() for k in ks)
        wbytes = sum(_streamed(ly) for ly in model.layers) + model.lm_head.numel() * model.lm_head.element_size()
        kvb = B * (S + args.new_tokens / 2.0) * tcfg.num_hidden_layers * 2 * tcfg.num_key_value_heads * tcfg.head_dim * 2
        decode_loop = {"ms_per_token": td / (args.new_tokens - 1) * 1e3, "steps": args.new_tokens - 1, "sequences": B,
                       "hipgraph": not args.no_graph,
                       "streamed_bytes_per_token": wbytes + kvb, "weight_bytes_per_token": wbytes, "kv_bytes_per_token": kvb,
                       "hbm_frac": (wbytes + kvb) / (td / (args.new_tokens - 1)) / 1e9 / PEAK_HBM_GBS,
                       "note": "host clock around generate_finish (decode steps only) of one extra batch after the timed region; "
                               "bytes = every layer's weights + lm_head once per token + K and V rows of every sequence"}
    # ---- roofline of the dominant kernel (bf16 tile GEMM), live HIP-event timing over the timed region ---------------
    def fold(ts):
        agg_ = {}
        for kind, flops, nbytes, e0, e1 in ts:
            a = agg_.setdefault(kind, [0.0, 0.0, 0.0, 0])
            a[0] += flops
            a[1] += nbytes
            a[2] += e0.elapsed_time(e1) * 1e-3
            a[3] += 1
        return agg_
    agg, pagg = fold(timers), fold(probe)
    step_s = elapsed / args.steps
    roof = None
    if "gemm_tile_bf16" in agg:
        fl, nb, sec, cnt = agg["gemm_tile_bf16"]
        roof = {"kernel": "gemm_bf16_pp_kernel (ViT qkv/proj/fc1/fc2, patch-embed, projector, Llama prefill qkv/o/gate-up/down)",
                "bound": "mfma", "achieved": fl / sec / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": fl / sec / 1e12 / PEAK_BF16_TFLOPS, "traffic": None,
                "launches": cnt, "avg_launch_us": sec / cnt * 1e6, "flop_per_launch": fl / cnt,
                "algorithmic_bytes_per_launch": nb / cnt, "time_share_of_step": sec / elapsed}
        # HBM-side bytes per launch come from separate rocprofv3 --pmc passes of this same command (FETCH_SIZE and
        # WRITE_SIZE cannot share a pass), folded by tools/pmc_summary.py and committed under profiles/ (newest round first)
        for rnd in ("r5", "r4", "r3", "r2", "r1"):
            pmc = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic.json")
            if args.workload == "single" and args.model == "gar_1b" and os.path.exists(pmc):
                try:
                    k = json.load(open(pmc))["kernels"]["gemm_bf16_pp_kernel"]
                    roof["traffic"] = k["traffic_bytes_per_launch"]
                    roof["traffic_source"] = f"profiles/{rnd}_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)"
                    break
                except Exception:
                    pass
    value = world * args.steps * B / elapsed
    mname = {"gar_1b": "GAR-1B", "gar_8b": "GAR-8B"}.get(args.model, args.model)
    if args.workload == "multi_region":
        metric, unit = f"prompts/sec (1024^2 img, 4 masks, relationship prompt, 64-tok answer) {mname}", "prompts/s"
        wl = f"{mname} bf16, synthetic 1024x1024 images, 4 masks/image multi-region relationship prompt " \
             f"(gar_relationship.py path), {args.new_tokens}-token greedy answer (BASELINE.json configs[2])"
    elif args.workload == "video":
        metric, unit = f"clips/sec (8-frame 1024^2 clip, per-frame mask, 64-tok caption) {mname}", "clips/s"
        wl = f"{mname} bf16 video replay, synthetic 8-frame 1024x1024 clips with a per-frame mask (VideoRefer-style), " \
             f"{args.new_tokens}-token greedy caption (BASELINE.json configs[4] shape on {world} GPU)"
    else:
        cfg_idx = {"gar_1b": "configs[1]", "gar_8b": "configs[3] shape"}.get(args.model, "parity config")
        metric, unit = f"regions/sec (1024^2 img, 1 mask, 64-tok caption) {mname}", "regions/s"
        wl = f"{mname} bf16, synthetic 1024x1024 images, 1 mask/region, {args.new_tokens}-token greedy caption " \
             f"(BASELINE.json {cfg_idx})"
    ids0 = batches[0]["input_ids"][0]
    plan_v, plan_l = model._plan_passes(B, tiles, S)
    plan_l_ = plan_l
    n_crop_rows = int(sum(int((ids0 == t).sum()) for t in (batches[0].get("video_frame_tokens") or cfg.crop_tokens_ids)))
    line = {"metric": metric, "value": value, "unit": unit,
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.data_type, "data": "synthetic",
            "config": {"workload": wl,
                       "regions_per_step_per_gpu": B, "passes": {"vision_tower_tiles": plan_v, "prefill_sequences": plan_l},
                       "decode": "eager launches" if args.no_graph else "one hipGraph replay per token",
                       "step_pipeline": ("decode loop of step i on a second HIP stream beside the prompt phase (vision tower, "
                                         "sequence assembly, prefill) of step i + 1, two KV-state slots; every step's prompt "
                                         "phase AND decode loop complete inside the timed region") if pipe is not None
                                        else "none: every step's decode loop runs behind its own prompt phase",
                       "tiles_per_region": tiles, "prefill_len": S,
                       "replayed_rows_per_region": n_crop_rows,
                       "new_tokens": args.new_tokens, "max_num_tiles": args.max_num_tiles,
                       "inputs": "resident in HBM" if args.preprocess == "resident" else
                                 "built per step from host images (device preprocessing inside the timed region)",
                       "distinct_samples": f"{n_distinct} distinct synthetic samples per rank ({args.pool} batches of {B}), "
                                           f"built with the device preprocessor before the timed region",
                       "validate": False,
                       "validate_note": "generate(validate=False): the reference's image-token count / span / missing-bbox "
                                        "checks run on the device inside the timed region and their flag is read after it "
                                        "(0 here); only the host syncs are skipped",
                       "eos": "disabled (exactly new_tokens tokens per region)",
                       "peak_device_memory_gib": rt.peak_mem_gib(device),
                       "weights": f"seeded synthetic {mname}", "parallelism": f"dp{world} (replica per GPU, RCCL weight "
                                                                              f"broadcast + caption gather)"},
            "roofline": roof,
            "parity_pins": parity_pins(),
            **dp.describe(),
            "caption_gather": {"rank0_ids_shape": gathered_shape, "per_step": True,
                               "region_partition": f"region i runs on rank i % {world} (rank r: r, r + {world}, r + {2 * world}, ...)"},
            "per_rank_ms_per_step": [x / args.steps * 1e3 for x in per_rank],
            "weight_broadcast": {"seconds": bcast_s, "bytes": bcast_bytes, "collectives": bcast_n,
                                 "note": "ONE in-place broadcast per dtype arena of the prepared weights (bytes = the arena) from rank 0 "
                                         "before the warm-up (RCCL over xGMI; no collective at N = 1)"}}
    # ---- every other kernel family of the step: achieved / peak / frac and its share of the step -------------------------
    other = line["roofline_other"] = {}

    def mfma_entry(name, a, note=None):
        fl, nb, sec, cnt = a
        other[name] = {"bound": "mfma", "achieved": fl / sec / 1e12, "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                       "frac": fl / sec / 1e12 / PEAK_BF16_TFLOPS, "launches": cnt, "avg_launch_us": sec / cnt * 1e6,
                       "time_share_of_step": sec / elapsed}
        if note:
            other[name]["note"] = note

    def hbm_entry(name, nb, sec, cnt, share, **extra):
        other[name] = {"bound": "hbm", "achieved": nb / sec / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                       "frac": nb / sec / 1e9 / PEAK_HBM_GBS, "launches": cnt, "avg_launch_us": sec / cnt * 1e6,
                       "time_share_of_step": share, **extra}

    if "attn_full" in agg:
        mfma_entry("attn_bf16_v2_kernel, ViT (non-causal, 1 + 1024 keys per tile and head)", agg["attn_full"],
                   "flops = 4 * head_dim * queries * keys per (tile, head)")
    if "attn_causal" in agg:
        mfma_entry("attn_bf16_v2_kernel, Llama prefill (causal GQA)", agg["attn_causal"],
                   "flops = 4 * head_dim * S (S + 1) / 2 per (sequence, query head): the causal triangle only")
    if "gemm_skinny_bf16" in agg:       # prompt phase: the first-token head and the pruned last layer's B-row o / gate-up / down
        fl, nb, sec, cnt = agg["gemm_skinny_bf16"]
        hbm_entry("skinny_mt_bf16_kernel, prompt phase (lm_head of the first token + the pruned last prefill layer's B-row GEMMs)",
                  nb, sec, cnt, sec / elapsed)
    if "attn_decode" in agg:            # prompt phase: the pruned last layer's single-row attention over S cache rows
        fl, nb, sec, cnt = agg["attn_decode"]
        t_ = cfg.mllm_config.text_config
        nb = sum(plan_l_) * t_.num_key_value_heads * S * t_.head_dim * 4.0 * args.steps
        hbm_entry("decode_attn_lds_kernel, prompt phase (last prompt row of the pruned last prefill layer)", nb, sec, cnt,
                  sec / elapsed)
    # HBM-bound RoI feature-replay pass (pool + embed/scatter + RoI replay), priced at SURVEY.md section 8d's ALGORITHMIC
    # bytes: the projector's grid rows read once + the sequence written once + the RoI cells (~90 MB per region at GAR-1B /
    # 1024^2). Since round 2 the pass IS that traffic: gar_pool_assemble pools on the way into the sequence and
    # gar_roi_replay_inplace reads the pooled rows back from it (round 1 wrote and re-read the pooled features: 128.9 MB).
    hb = [agg[k] for k in ("pool_assemble", "roi_replay") if k in agg]
    if hb:
        nb, sec = sum(a[1] for a in hb), sum(a[2] for a in hb)
        hbm_entry("RoI feature-replay pass (pool_assemble + roi_replay_inplace)", nb, sec, sum(a[3] for a in hb), sec / elapsed,
                  algorithmic_bytes_per_region=nb / (args.steps * B), us_per_region=sec / (args.steps * B) * 1e6)
    if "roi_replay" in agg:
        fl, nb, sec, cnt = agg["roi_replay"]
        hbm_entry("roi_replay_inplace_kernel alone (one launch per prefill pass; ~1 MB per crop token: launch / latency bound)",
                  nb, sec, cnt, sec / elapsed, bytes_per_launch=nb / cnt)
    # decode kernels (inside the hipGraph during the timed region): per-step time from the eager probe x (new_tokens - 1) steps
    if pagg:
        t_ = cfg.mllm_config.text_config
        nsteps_probe = args.decode_probe_steps
        dec_steps = args.new_tokens - 1
        probe_note = (f"timed on {nsteps_probe} eager decode steps of the same batch after the timed region (inside it these "
                      f"kernels replay from the hipGraph); time_share_of_step = per-step time x {dec_steps} steps / step time")
        if "decode:attn_decode" in pagg:
            fl, nb, sec, cnt = pagg["decode:attn_decode"]
            nb = cnt * B * t_.num_key_value_heads * (S + 1 + nsteps_probe / 2.0) * t_.head_dim * 4.0    # K + V rows of every sequence
            hbm_entry("decode_attn_lds_kernel, decode steps (KV cache streamed once per step and layer)", nb, sec, cnt,
                      sec / nsteps_probe * dec_steps / step_s, us_per_decode_step=sec / nsteps_probe * 1e6, note=probe_note)
        if "decode:gemm_skinny_bf16" in pagg:
            fl, nb, sec, cnt = pagg["decode:gemm_skinny_bf16"]
            hbm_entry("skinny_mt_bf16_kernel, decode steps (qkv / o / gate-up / down GEMVs of every layer + lm_head: weights "
                      "streamed once per step for all B rows)", nb, sec, cnt, sec / nsteps_probe * dec_steps / step_s,
                      us_per_decode_step=sec / nsteps_probe * 1e6, weight_bytes_per_decode_step=nb / nsteps_probe, note=probe_note)
        if "decode:splitk_reduce" in pagg:
            fl, nb, sec, cnt = pagg["decode:splitk_reduce"]
            hbm_entry("splitk_res_rms_kernel, decode steps (split-K reduce + residual (+ final RMSNorm))", nb, sec, cnt,
                      sec / nsteps_probe * dec_steps / step_s, us_per_decode_step=sec / nsteps_probe * 1e6, note=probe_note)
        tot = sum(a[2] for a in pagg.values())
        line["decode_step_probe"] = {"eager_us_per_step_sum_of_kernels": tot / nsteps_probe * 1e6, "steps": nsteps_probe,
                                     "kernels_per_step": sum(a[3] for a in pagg.values()) / nsteps_probe}
    if decode_loop:         # measured whether or not the eager probe ran (--decode-probe-steps 0 must not drop it: ADVICE r4)
        line["decode_loop"] = decode_loop
    covered = sum(v.get("time_share_of_step", 0.0) for k, v in other.items() if not k.startswith("roi_replay_inplace_kernel alone"))
    line["time_share_covered"] = (roof["time_share_of_step"] if roof else 0.0) + covered
    if not args.no_cpu_baseline and world == 1 and args.workload == "video":
        line["cpu_baseline"] = {"value": None, "note": "the CPU oracle leg is wired for the image workloads (run "
                                                       "--workload single for the headline line's baseline)"}
    elif not args.no_cpu_baseline and world == 1:        # reported on rank 0 at N=1 only
        try:
            line["cpu_baseline"] = cpu_baseline(cfg, W, one, args.new_tokens, args.cpu_threads)
        except Exception as e:  # the baseline is a reported number, never a reason to lose the GPU result
            line["cpu_baseline"] = {"value": None, "error": repr(e)}
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
