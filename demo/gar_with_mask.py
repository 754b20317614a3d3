#!/usr/bin/env python
"""Region captioning from an image + a binary mask on the MI355X path — CLI counterpart of the reference's
demo/gar_with_mask.py (same flags: --model_name_or_path --image_path --mask_path --data_type --seed)."""
import numpy as np
from PIL import Image

from _common import base_parser, generation_config, load


def main():
    ap = base_parser("Inference demo of Grasp Any Region models (MI355X-native path).")
    ap.add_argument("--image_path", required=True)
    ap.add_argument("--mask_path", required=True)
    args = ap.parse_args()
    model, processor, dtype = load(args)
    from evaluation.eval_dataset import SingleRegionCaptionDataset
    img = Image.open(args.image_path)
    mask = np.array(Image.open(args.mask_path).convert("L")).astype(bool)
    prompt_number = model.config.prompt_numbers
    prompt_tokens = [f"<Prompt{i}>" for i in range(prompt_number)] + ["<NO_Prompt>"]
    dataset = SingleRegionCaptionDataset(image=img, mask=mask, processor=processor, prompt_number=prompt_number,
                                         visual_prompt_tokens=prompt_tokens, data_dtype=dtype, device=args.device)
    data_sample = dataset[0]
    out = model.generate(**data_sample, generation_config=generation_config(args, processor), return_dict=True)
    print(processor.tokenizer.decode(out.sequences[0], skip_special_tokens=True).strip())


if __name__ == "__main__":
    main()
