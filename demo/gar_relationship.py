#!/usr/bin/env python
"""Multi-region relationship question on the MI355X path — CLI counterpart of the reference's demo/gar_relationship.py
(--image_path, --mask_paths ..., --question_str with <PromptK> tokens)."""
import numpy as np
from PIL import Image

from _common import base_parser, generation_config, load


def main():
    ap = base_parser("Multi-region inference demo of Grasp Any Region models (MI355X-native path).")
    ap.add_argument("--image_path", required=True)
    ap.add_argument("--mask_paths", nargs="+", required=True)
    ap.add_argument("--question_str", required=True)
    args = ap.parse_args()
    model, processor, dtype = load(args)
    from evaluation.eval_dataset import MultiRegionDataset
    img = Image.open(args.image_path)
    masks = [np.array(Image.open(p).convert("L")).astype(bool) for p in args.mask_paths]
    prompt_number = model.config.prompt_numbers
    prompt_tokens = [f"<Prompt{i}>" for i in range(prompt_number)] + ["<NO_Prompt>"]
    question = args.question_str + "\nAnswer with the correct option's letter directly."
    dataset = MultiRegionDataset(image=img, masks=masks, question_str=question, processor=processor,
                                 prompt_number=prompt_number, visual_prompt_tokens=prompt_tokens, data_dtype=dtype,
                                 device=args.device)
    data_sample = dataset[0]
    out = model.generate(**data_sample, generation_config=generation_config(args, processor), return_dict=True)
    print(processor.tokenizer.decode(out.sequences[0], skip_special_tokens=True).strip())


if __name__ == "__main__":
    main()
