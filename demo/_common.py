"""Shared by the demo CLIs: model / processor construction with the reference's flags plus --device,
--max_num_tiles and --synthetic_weights (no checkpoint is reachable offline)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

from gar_amd.bench_loops import DATA_TYPE_CHOICES, TORCH_DTYPE_MAP, resolve_data_type  # noqa: E402,F401


def base_parser(description):
    ap = argparse.ArgumentParser(description=description)
    ap.add_argument("--model_name_or_path", default="HaochenWang/GAR-1B",
                    help="checkpoint directory (config.json + *.safetensors [+ tokenizer]); with --synthetic_weights a "
                         "size name: gar_1b | gar_8b | tiny")
    ap.add_argument("--data_type", choices=DATA_TYPE_CHOICES, default="bf16",
                    help="bf16 | fp16 | fp32 (parity mode)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--max_num_tiles", type=int, default=16)
    ap.add_argument("--max_new_tokens", type=int, default=1024)
    ap.add_argument("--host_preprocessing", action="store_true",
                    help="resize / tile / normalise on the CPU (torch bicubic) instead of the device kernels")
    ap.add_argument("--synthetic_weights", action="store_true",
                    help="seeded random weights of the named size + the stub tokenizer (plumbing / benchmarking)")
    return ap


def load(args):
    from gar_amd import GARConfig
    from gar_amd.modeling_gar import GARModel
    from gar_amd.processing import GARProcessor
    dtype = resolve_data_type(args.data_type)
    torch.manual_seed(args.seed)
    if args.synthetic_weights:
        name = args.model_name_or_path if args.model_name_or_path in ("gar_1b", "gar_8b", "tiny") else "gar_1b"
        cfg = getattr(GARConfig, name)()
        model = GARModel.from_synthetic(cfg, args.seed, dtype, args.device)
        processor = GARProcessor.from_config(cfg, max_num_tiles=args.max_num_tiles)
    else:
        model = GARModel.from_pretrained(args.model_name_or_path, dtype, args.device)
        processor = GARProcessor.from_pretrained(args.model_name_or_path, model.config, args.max_num_tiles)
    if not args.host_preprocessing:
        processor.use_gpu_preprocessing(args.device, dtype)
    return model.eval(), processor, dtype


def generation_config(args, processor):
    # transformers.GenerationConfig(max_new_tokens, do_sample=False, eos, pad) of the reference, as a plain dict
    return dict(max_new_tokens=args.max_new_tokens, do_sample=False, eos_token_id=processor.tokenizer.eos_token_id,
                pad_token_id=processor.tokenizer.pad_token_id)
