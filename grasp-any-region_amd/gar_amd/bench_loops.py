"""Benchmark inference loops: counterparts of the reference's evaluation/GAR-Bench/inference.py
and evaluation/DLC-Bench/inference.py on the MI355X path. Same flags, question templates, output JSON formats and the
VQA exact-match accuracy print-out; masks are decoded with ``gar_amd.rle`` instead of pycocotools. Items are sharded
round-robin over the ranks of a torchrun launch (one replica per GPU) and gathered on rank 0."""
from __future__ import annotations

import argparse
import ast
import json
import os

import torch
from PIL import Image

from . import dp, rle

TORCH_DTYPE_MAP = dict(bf16=torch.bfloat16, fp16=torch.float16, fp32=torch.float32)
DATA_TYPE_CHOICES = ["fp16", "bf16", "fp32"]   # the reference CLIs' choices (demo/gar_with_mask.py:44)


def resolve_data_type(name):
    """--data_type of the reference CLIs (demo/gar_with_mask.py:41-45): bf16 (default, what the released checkpoints are
    stored in), fp16 (the twin library libgar_hip_f16.so: the same kernels with IEEE binary16 as the 16-bit element type)
    or fp32 (parity mode)."""
    return TORCH_DTYPE_MAP[name]


def base_parser(description, default_model, default_cache, default_images):
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("--model_name_or_path", default=default_model,
                    help="checkpoint directory; with --synthetic_weights a size name: gar_1b | gar_8b | tiny")
    ap.add_argument("--cache_name", type=str, default=default_cache, help="cache name for saving results")
    ap.add_argument("--anno_file", required=True, help="annotation file path")
    ap.add_argument("--image_folder", default=default_images, help="the folder of images")
    ap.add_argument("--data_type", choices=DATA_TYPE_CHOICES, default="bf16",
                    help="bf16 | fp16 | fp32 (parity mode)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default=None, help="default: cuda:LOCAL_RANK")
    ap.add_argument("--max_num_tiles", type=int, default=16)
    ap.add_argument("--max_new_tokens", type=int, default=1024)
    ap.add_argument("--output_dir", default=None, help="default: the reference's model_outputs directory")
    ap.add_argument("--limit", type=int, default=0, help="only the first N items (smoke runs)")
    ap.add_argument("--host_preprocessing", action="store_true")
    ap.add_argument("--batch_size", type=int, default=1,
                    help="items per generate call. 1 (default) = one item per call like the reference's loops. n > 1: prompts of "
                         "different lengths go in as ONE left-padded batch with an attention_mask (items with the same tile count "
                         "are grouped) — mathematically the same function, but in bf16 the batch size selects kernels (skinny vs "
                         "tile GEMM, split-K, folded norms), so a caption can differ from the per-item run at near-tied steps; the "
                         "mode a file was produced in is recorded next to it (<cache>.meta.json)")
    ap.add_argument("--continuous", action="store_true",
                    help="with --batch_size n > 1: keep n decode rows full instead of running static batches of n — a row whose "
                         "caption hits EOS is retired and the next queued item is admitted into it inside the running decode loop "
                         "(gar_amd/continuous.py), so a batch no longer pays for its longest caption in every row")
    ap.add_argument("--synthetic_weights", action="store_true")
    return ap


def load(args):
    from .configuration_gar import GARConfig
    from .modeling_gar import GARModel
    from .processing import GARProcessor
    dtype = resolve_data_type(args.data_type)
    rank, local, world = dp.init_distributed()
    device = args.device or f"cuda:{local}"
    torch.cuda.set_device(torch.device(device))
    torch.manual_seed(args.seed)
    if args.synthetic_weights:
        name = args.model_name_or_path if args.model_name_or_path in ("gar_1b", "gar_8b", "tiny") else "gar_1b"
        cfg = getattr(GARConfig, name)()
        model = GARModel.from_synthetic(cfg, args.seed, dtype, device)
        processor = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles)
    else:
        model = GARModel.from_pretrained(args.model_name_or_path, dtype, device)
        processor = GARProcessor.from_pretrained(args.model_name_or_path, model.config, args.max_num_tiles)
    if not args.host_preprocessing:
        processor.use_gpu_preprocessing(device, dtype)
    args._model = model                     # write_run_meta reads the fused-path switches off it
    return model.eval(), processor, dtype, device, rank, world


def write_run_meta(path, args, model):
    """<outputs>.meta.json beside a loop's output file: how the captions were produced (the output file keeps the reference's
    record format). In bf16, batch size and pass sizes select kernels, so two runs of the same items can differ at near-tied
    greedy steps ("Batching and caption-level drift")."""
    meta = {"batch_size": int(getattr(args, "batch_size", 1) or 1), "continuous": bool(getattr(args, "continuous", False)),
            "data_type": args.data_type,
            "max_num_tiles": args.max_num_tiles, "max_new_tokens": args.max_new_tokens,
            "synthetic_weights": bool(args.synthetic_weights), "host_preprocessing": bool(args.host_preprocessing),
            "fused_paths": {k: bool(getattr(model, k)) for k in ("FOLD_NORMS", "LLM_QKV_EPILOGUE", "DECODE_GU_NORM_FOLDED",
                                                                 "DECODE_ATTN_TAKES_QKV", "VIT_CLS_KEY_FOLD",
                                                                 "PRUNE_LAST_PREFILL_LAYER") if hasattr(model, k)},
            "abi_version": hip_abi_version()}
    json.dump(meta, open(os.path.splitext(path)[0] + ".meta.json", "w"), indent=2)


def hip_abi_version():
    from . import hip
    return hip.ABI_VERSION


def _generate(model, processor, sample, args, skip_special_tokens):
    out = model.generate(**sample, generation_config=dict(
        max_new_tokens=args.max_new_tokens, do_sample=False, eos_token_id=processor.tokenizer.eos_token_id,
        pad_token_id=processor.tokenizer.pad_token_id), return_dict=True)
    return processor.tokenizer.decode(out.sequences[0], skip_special_tokens=skip_special_tokens).strip()


class Batcher:
    """Continuous batching of the benchmark items: ``add`` queues a sample with the function that
    turns its decoded text into the loop's record; samples with the same tile count (and the same video frame tokens)
    are run ``batch_size`` at a time as ONE ``generate`` — their prompts right-aligned in a left-padded ``input_ids``
    with the ``attention_mask`` the reference forwards to HF's generate (modeling_gar.py:418-426) — so the decode weight
    stream is shared by the whole batch instead of being paid per item. A row's text is cut at its own first EOS, which is
    what a one-item call returns. ``generate_calls`` / ``items`` count what was run (tests)."""

    def __init__(self, model, processor, args, skip_special_tokens):
        self.model, self.processor, self.args, self.skip = model, processor, args, skip_special_tokens
        self.bs = max(1, int(getattr(args, "batch_size", 1) or 1))
        self.groups = {}
        self.results = []
        self.generate_calls = 0
        self.items = 0

    def add(self, idx, sample, finish):
        if self.bs == 1:
            self._emit(idx, _generate(self.model, self.processor, sample, self.args, self.skip), finish)
            self.generate_calls += 1
            self.items += 1
            return
        key = (int(sample["pixel_values"].shape[0]), tuple(sample.get("video_frame_tokens") or ()),
               bool(sample.get("feature_replay_video")))
        g = self.groups.setdefault(key, [])
        g.append((idx, sample, finish))
        if len(g) >= self.bs:
            self._run(self.groups.pop(key))

    def flush(self):
        for key in list(self.groups):
            self._run(self.groups.pop(key))
        print(f"[batcher] {self.items} items in {self.generate_calls} generate calls (batch_size {self.bs})", flush=True)
        return self.results

    def _emit(self, idx, text, finish):
        print(text, flush=True)
        self.results.append((idx, finish(text)))

    def _run(self, group):
        tk = self.processor.tokenizer
        eos = tk.eos_token_id
        eos_set = set(eos) if isinstance(eos, (list, tuple)) else {eos}
        pad = tk.pad_token_id if tk.pad_token_id is not None else next(iter(eos_set))
        samples = [s for _, s, _ in group]
        S = max(int(s["input_ids"].shape[1]) for s in samples)
        dev = samples[0]["input_ids"].device
        ids = torch.full((len(samples), S), int(pad), dtype=torch.int64, device=dev)
        mask = torch.zeros((len(samples), S), dtype=torch.int64, device=dev)
        for b, s in enumerate(samples):
            n = int(s["input_ids"].shape[1])
            ids[b, S - n:] = s["input_ids"][0]
            mask[b, S - n:] = 1
        batch = dict(input_ids=ids, attention_mask=mask,
                     pixel_values=torch.cat([s["pixel_values"] for s in samples]),
                     global_mask_values=torch.cat([s["global_mask_values"] for s in samples]),
                     bboxes=[s["bboxes"][0] for s in samples])
        if samples[0].get("aspect_ratios") is not None:
            batch["aspect_ratios"] = torch.cat([s["aspect_ratios"] for s in samples])
        if samples[0].get("feature_replay_video"):
            batch.update(feature_replay_video=True, video_frame_tokens=samples[0]["video_frame_tokens"])
        out = self.model.generate(**batch, generation_config=dict(
            max_new_tokens=self.args.max_new_tokens, do_sample=False, eos_token_id=eos, pad_token_id=pad), return_dict=True)
        self.generate_calls += 1
        self.items += len(samples)
        rows = out.sequences.tolist()
        for (idx, _, finish), row in zip(group, rows):
            cut = next((j + 1 for j, t in enumerate(row) if t in eos_set), len(row))     # its own EOS, inclusive
            self._emit(idx, tk.decode(row[:cut], skip_special_tokens=self.skip).strip(), finish)


class ContinuousRunner:
    """The Batcher's interface (``add`` / ``flush`` / ``results``) over :class:`gar_amd.continuous.ContinuousBatcher`: the items
    are a QUEUE in front of ``batch_size`` decode rows that are refilled as captions end (--continuous)."""

    def __init__(self, model, processor, args, skip_special_tokens):
        from .continuous import ContinuousBatcher
        self.processor, self.skip = processor, skip_special_tokens
        tk = processor.tokenizer
        self.cb = ContinuousBatcher(model, slots=max(2, int(args.batch_size)), max_new_tokens=args.max_new_tokens,
                                    eos_token_id=tk.eos_token_id, poll_every=getattr(args, "poll_every", 8))
        self.pending = {}
        self.results = []
        self.items = 0

    def _drain(self):
        tk = self.processor.tokenizer
        for ticket, toks in self.cb.pop_finished():
            idx, finish = self.pending.pop(ticket)
            text = tk.decode(toks, skip_special_tokens=self.skip).strip()
            print(text, flush=True)
            self.results.append((idx, finish(text)))

    def add(self, idx, sample, finish):
        self.pending[self.cb.submit(sample)] = (idx, finish)
        self.items += 1
        self.cb.pump()
        self._drain()

    def flush(self):
        self.cb.flush()
        self._drain()
        st = self.cb.stats
        print(f"[continuous] {self.items} items, {self.cb.B} rows: {st['prompt_passes']} prompt passes, {st['decode_steps']} decode "
              f"steps, row occupancy {st['live_row_steps'] / max(1, st['row_steps']):.2f}, {st['rebases']} re-bases", flush=True)
        return self.results


def make_runner(model, processor, args, skip_special_tokens):
    if getattr(args, "continuous", False) and int(getattr(args, "batch_size", 1) or 1) > 1:
        return ContinuousRunner(model, processor, args, skip_special_tokens)
    return Batcher(model, processor, args, skip_special_tokens)


def _gather(local, rank, world):
    """[(index, value)] from every rank -> index-sorted list on rank 0 (None elsewhere)."""
    if world == 1:
        return [v for _, v in sorted(local, key=lambda t: t[0])]
    import torch.distributed as dist
    allr = [None] * world
    dist.all_gather_object(allr, local)
    if rank != 0:
        return None
    return [v for _, v in sorted((t for part in allr for t in part), key=lambda t: t[0])]


def gar_bench_question(item, mode):
    """evaluation/GAR-Bench/inference.py:124-135"""
    if mode == "vqa":
        q = f"Question: {item['question']}\nOptions:"
        for op in item["choices"]:
            q += f"\n{op}"
        return q + "\nAnswer with the correct option's letter directly."
    if mode == "simple":
        return item["question"]
    if mode == "detailed":
        return "Describe <Prompt0> in detail, including the relationship with <Prompt1>."
    raise NotImplementedError(mode)


def run_gar_bench(argv=None):
    from .eval_dataset import MultiRegionDataset
    ap = base_parser("Inference of Grasp Any Region models on GAR-Bench (MI355X-native path).", "HaochenWang/GAR-8B",
                     "gar_8b", "evaluation/GAR-Bench/annotations")
    ap.add_argument("--mode", choices=["vqa", "simple", "detailed"], required=True, help="mode to build questions")
    args = ap.parse_args(argv)
    model, processor, dtype, device, rank, world = load(args)
    data = json.load(open(args.anno_file))
    if args.limit:
        data = data[:args.limit]
    prompt_number = model.config.prompt_numbers
    prompt_tokens = [f"<Prompt{i}>" for i in range(prompt_number)] + ["<NO_Prompt>"]
    runner = make_runner(model, processor, args, skip_special_tokens=False)

    def record(item):
        def finish(text):
            if text.endswith("<|eot_id|>"):
                text = text.replace("<|eot_id|>", "")
            return dict(item, model_output=text)
        return finish
    for idx in dp.shard_indices(len(data), rank, world):
        item = data[idx]
        img = Image.open(os.path.join(args.image_folder, item["image"]))
        masks = [(rle.decode(r) * 255).astype("uint8") for r in item["mask_rles"]]
        ds = MultiRegionDataset(image=img, masks=masks, question_str=gar_bench_question(item, args.mode),
                                processor=processor, prompt_number=prompt_number, visual_prompt_tokens=prompt_tokens,
                                data_dtype=dtype, device=device)
        runner.add(idx, ds[0], record(item))
    outputs = _gather(runner.flush(), rank, world)
    if outputs is None:
        return None
    cache = f"{args.cache_name}_{args.mode}"
    print(f"Cache name: {cache}")
    out_dir = args.output_dir or "evaluation/GAR-Bench/model_outputs"
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{cache}.json")
    json.dump(outputs, open(path, "w"), indent=4, ensure_ascii=False)
    write_run_meta(path, args, args._model)
    if args.mode == "vqa":      # exact-match accuracy per category and overall (:185-203)
        for cat in sorted(set(x["type"] for x in outputs)):
            res = [x for x in outputs if x["type"] == cat]
            ok = len([x for x in res if x["model_output"].lower() == x["answer"].lower()])
            print(f"{cat}: [{ok}/{len(res)}]={round(ok / len(res) * 100, 1)}")
        ok = len([x for x in outputs if x["model_output"].lower() == x["answer"].lower()])
        print(f"=> overall: [{ok}/{len(outputs)}]={round(ok / len(outputs) * 100, 1)}")
    return path


def run_dlc_bench(argv=None):
    """COCO-style annotation file (an Objects365 subset): one caption per annotation, keyed by annotation id
    (evaluation/DLC-Bench/inference.py:108-166)."""
    from .eval_dataset import SingleRegionCaptionDataset
    ap = base_parser("Inference of Grasp Any Region models on DLC-Bench (MI355X-native path).", "HaochenWang/GAR-8B",
                     "gar_8b", "evaluation/DLC-Bench/annotations")
    args = ap.parse_args(argv)
    model, processor, dtype, device, rank, world = load(args)
    coco = json.load(open(args.anno_file))
    imgs = {str(i["id"]): i for i in coco["images"]}
    anns = coco["annotations"]          # the reference walks images, then each image's annotations: same set
    order = sorted(range(len(anns)), key=lambda k: (list(imgs).index(str(anns[k]["image_id"])), k))
    if args.limit:
        order = order[:args.limit]
    prompt_number = model.config.prompt_numbers
    prompt_tokens = [f"<Prompt{i}>" for i in range(prompt_number)] + ["<NO_Prompt>"]
    runner = make_runner(model, processor, args, skip_special_tokens=True)
    for j in dp.shard_indices(len(order), rank, world):
        a = anns[order[j]]
        seg = ast.literal_eval(a["segmentation"]) if isinstance(a["segmentation"], str) else a["segmentation"]
        mask = rle.decode(seg)
        info = imgs[str(a["image_id"])]
        img = Image.open(os.path.join(args.image_folder, "images", info["file_name"]))
        ds = SingleRegionCaptionDataset(image=img, mask=mask, processor=processor, prompt_number=prompt_number,
                                        visual_prompt_tokens=prompt_tokens, data_dtype=dtype, device=device)
        runner.add(j, ds[0], (lambda aid: lambda text: (aid, text))(a["id"]))
    outputs = _gather(runner.flush(), rank, world)
    if outputs is None:
        return None
    out_dir = args.output_dir or "evaluation/DLC-Bench/model_outputs"
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{args.cache_name}.json")
    json.dump({k: v for k, v in outputs}, open(path, "w"), indent=4, ensure_ascii=False)
    write_run_meta(path, args, args._model)
    print(f"Cache name: {args.cache_name}")
    return path


def _single_region_loop(args, items, get_image, get_mask, make_record):
    """Shared body of the Ferret-Bench / MDVP-Bench loops: one SingleRegionCaptionDataset sample per item."""
    from .eval_dataset import SingleRegionCaptionDataset
    model, processor, dtype, device, rank, world = load(args)
    if args.limit:
        items = items[:args.limit]
    prompt_number = model.config.prompt_numbers
    prompt_tokens = [f"<Prompt{i}>" for i in range(prompt_number)] + ["<NO_Prompt>"]
    runner = make_runner(model, processor, args, skip_special_tokens=True)
    for idx in dp.shard_indices(len(items), rank, world):
        item = items[idx]
        image_path, img = get_image(item)
        mask = get_mask(item, img)
        ds = SingleRegionCaptionDataset(image=img, mask=mask, processor=processor, prompt_number=prompt_number,
                                        visual_prompt_tokens=prompt_tokens, data_dtype=dtype, device=device)
        runner.add(idx, ds[0], (lambda it, p: lambda text: make_record(it, p, text))(item, image_path))
    return _gather(runner.flush(), rank, world)


def run_ferret_bench(argv=None):
    """evaluation/Ferret-Bench/inference.py:67-163 — polygon (frPyObjects + merge) or RLE segmentations, records
    {image_path, annotation, caption}."""
    ap = base_parser("Inference of Grasp Any Region models on Ferret-Bench (MI355X-native path).", "HaochenWang/GAR-8B",
                     "gar_8b", "evaluation/Ferret-Bench/annotations")
    args = ap.parse_args(argv)
    data = json.load(open(args.anno_file))

    def get_image(item):
        p = os.path.join(args.image_folder, item["image"])
        return p, Image.open(p).convert("RGB")

    def get_mask(item, img):
        seg = item["annotation"]["segmentation"]
        seg = ast.literal_eval(seg) if isinstance(seg, str) else seg
        w, h = img.size
        m = rle.from_polygons(seg, h, w) if isinstance(seg, list) else rle.decode(seg)
        return (m * 255).astype("uint8")

    outputs = _single_region_loop(args, data, get_image, get_mask,
                                  lambda item, p, text: {"image_path": p, "annotation": item["annotation"], "caption": text})
    if outputs is None:
        return None
    out_dir = args.output_dir or "evaluation/Ferret-Bench/model_outputs"
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{args.cache_name}.json")
    json.dump(outputs, open(path, "w"), indent=4, ensure_ascii=False)
    write_run_meta(path, args, args._model)
    print(f"Cache name: {args.cache_name}")
    return path


def run_mdvp_bench(argv=None):
    """evaluation/MDVP-Bench/inference.py:108-161 — `mask_rle` entries, default max_num_tiles 8, records
    {image_path, caption, gt}."""
    ap = base_parser("Inference of Grasp Any Region models on MDVP-Bench (MI355X-native path).", "HaochenWang/GAR-1B",
                     "gar_1b", "evaluation/MDVP-Bench/data")
    ap.set_defaults(max_num_tiles=8, anno_file="evaluation/MDVP-Bench/annotations/mdvp_caption_mask.json")
    for a in ap._actions:
        if a.dest == "anno_file":
            a.required = False
    args = ap.parse_args(argv)
    data = json.load(open(args.anno_file))

    def get_image(item):
        p = os.path.join(args.image_folder, item["image_path"])
        return p, Image.open(p).convert("RGB")

    outputs = _single_region_loop(args, data, get_image,
                                  lambda item, img: (rle.decode(item["mask_rle"]) * 255).astype("uint8"),
                                  lambda item, p, text: {"image_path": p, "caption": text, "gt": item["caption"]})
    if outputs is None:
        return None
    out_dir = args.output_dir or "evaluation/MDVP-Bench/model_outputs"
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{args.cache_name}.json")
    json.dump(outputs, open(path, "w"), indent=4, ensure_ascii=False)
    write_run_meta(path, args, args._model)
    print(f"Cache name: {args.cache_name}")
    return path


def sample_frame_indices(n_frames: int, n_keep: int = 8):
    """``n_keep`` indices spread uniformly over ``n_frames`` (first and last included), all of them when fewer."""
    if n_frames <= n_keep:
        return list(range(n_frames))
    return [round(i * (n_frames - 1) / (n_keep - 1)) for i in range(n_keep)]


def run_video_refer(argv=None):
    """VideoRefer-style driver for the video replay path (A13, modeling_perception_lm.py:765-852). The reference ships
    the model-side code of that path but no caller, so the annotation layout is this
    repo's, modelled on VideoRefer-Bench's mask annotations:

        [{"id": ..., "video": "<directory of frame images>" | "frames": ["f0.jpg", ...],
          "annotation": [{"<frame index>": {"segmentation": <COCO RLE | polygons>}}, ...]   # one object, per-frame masks
          | "masks": {"<frame index>": <RLE | polygons>},
          "question": "..." (optional)}]

    Up to 8 annotated frames per clip (uniformly sampled when there are more — the path has five crop tokens plus the
    following reserved ids, 8 frames at most); each frame becomes ONE 448-px tile with its own mask and crop token
    (``VideoRegionCaptionDataset``). Output: [{"id", "video", "frames", "caption"}] in item order."""
    from .eval_dataset import VideoRegionCaptionDataset
    ap = base_parser("Video region captioning with Grasp Any Region models (VideoRefer-style, MI355X-native path).",
                     "HaochenWang/GAR-8B", "gar_8b_video", "evaluation/VideoRefer-Bench/videos")
    ap.set_defaults(max_num_tiles=8)
    ap.add_argument("--num_frames", type=int, default=8)
    args = ap.parse_args(argv)
    if not 1 <= args.num_frames <= 8:
        raise SystemExit("--num_frames must be in 1..8 (the video replay path has 8 frame tokens)")
    model, processor, dtype, device, rank, world = load(args)
    data = json.load(open(args.anno_file))
    if args.limit:
        data = data[:args.limit]
    exts = (".jpg", ".jpeg", ".png", ".bmp", ".webp")
    runner = make_runner(model, processor, args, skip_special_tokens=True)
    for idx in dp.shard_indices(len(data), rank, world):
        item = data[idx]
        if "frames" in item:
            names = [os.path.join(args.image_folder, f) for f in item["frames"]]
        else:
            vdir = os.path.join(args.image_folder, item["video"])
            names = [os.path.join(vdir, f) for f in sorted(os.listdir(vdir)) if f.lower().endswith(exts)]
        per_frame = {}
        if "masks" in item:
            per_frame = {int(k): v for k, v in item["masks"].items()}
        else:
            for obj in item["annotation"][:1]:                      # one object per item (one caption)
                per_frame = {int(k): v["segmentation"] for k, v in obj.items()}
        annotated = sorted(k for k in per_frame if 0 <= k < len(names))
        if not annotated:
            raise ValueError(f"item {item.get('id', idx)}: no annotated frame inside the clip")
        keep = [annotated[i] for i in sample_frame_indices(len(annotated), args.num_frames)]
        frames, masks = [], []
        for k in keep:
            img = Image.open(names[k]).convert("RGB")
            seg = per_frame[k]
            seg = ast.literal_eval(seg) if isinstance(seg, str) else seg
            m = rle.from_polygons(seg, img.height, img.width) if isinstance(seg, list) else rle.decode(seg)
            frames.append(img)
            masks.append(m.astype(bool))
        kw = {"question": item["question"]} if item.get("question") else {}
        ds = VideoRegionCaptionDataset(frames, masks, processor, data_dtype=dtype, device=device, **kw)
        runner.add(idx, ds[0], (lambda it, i, kp: lambda text: {"id": it.get("id", i), "video": it.get("video"), "frames": kp,
                                                                  "caption": text})(item, idx, keep))
    outputs = _gather(runner.flush(), rank, world)
    if outputs is None:
        return None
    out_dir = args.output_dir or "evaluation/VideoRefer-Bench/model_outputs"
    os.makedirs(out_dir, exist_ok=True)
    path = os.path.join(out_dir, f"{args.cache_name}.json")
    json.dump(outputs, open(path, "w"), indent=4, ensure_ascii=False)
    write_run_meta(path, args, args._model)
    print(f"Cache name: {args.cache_name}")
    return path
