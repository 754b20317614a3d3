"""debug: multi-region sample through host vs device preprocessing — which tensors differ, and is generate deterministic."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
    sys.path.insert(0, p)
import numpy as np, torch
from gar_amd import GARConfig
from gar_amd.eval_dataset import MultiRegionDataset, SingleRegionCaptionDataset
from gar_amd.modeling_gar import GARModel
from gar_amd.processing import GARProcessor
from gar_amd.synthetic import synthetic_disjoint_masks, synthetic_image, synthetic_mask
cfg = GARConfig.tiny()
img = synthetic_image(11, 300, 220)
masks = [m.astype(np.uint8) * 255 for m in synthetic_disjoint_masks(11, 3, 300, 220)]
q = "What is the relationship between <Prompt0>, <Prompt1> and <Prompt2>?"
ph = GARProcessor.from_config(cfg, max_num_tiles=4)
pd = GARProcessor.from_config(cfg, max_num_tiles=4).use_gpu_preprocessing("cuda:0", torch.float32)
pt = [f"<Prompt{i}>" for i in range(cfg.prompt_numbers)] + ["<NO_Prompt>"]
a = MultiRegionDataset(image=img, masks=masks, question_str=q, processor=ph, prompt_number=cfg.prompt_numbers, visual_prompt_tokens=pt, data_dtype=torch.float32, device="cuda:0")[0]
b = MultiRegionDataset(image=img, masks=masks, question_str=q, processor=pd, prompt_number=cfg.prompt_numbers, visual_prompt_tokens=pt, data_dtype=torch.float32, device="cuda:0")[0]
for k in a:
    if torch.is_tensor(a[k]):
        same = a[k].shape == b[k].shape and torch.equal(a[k].float().cpu(), b[k].float().cpu())
        print(k, tuple(a[k].shape), tuple(b[k].shape), a[k].dtype, b[k].dtype, "contig", a[k].is_contiguous(), b[k].is_contiguous(), "SAME" if same else "DIFF")
        if not same and a[k].shape == b[k].shape:
            d = (a[k].float().cpu() - b[k].float().cpu()).abs()
            print("   ndiff", int((d > 0).sum()), "max", float(d.max()), "uniq a", torch.unique(a[k].float().cpu())[:10].tolist(), "uniq b", torch.unique(b[k].float().cpu())[:10].tolist())
    else:
        print(k, a[k] == b[k], a[k] if k == "bboxes" else "")
m = GARModel.from_synthetic(cfg, 0, torch.float32)
for name, s in (("host", a), ("dev", b), ("host", a), ("dev", b)):
    o = m.generate(**s, max_new_tokens=8, return_logits=True)
    print(name, o.sequences[0].tolist(), float(o.logits[0, 0].abs().max()), float(o.logits[0, 0].sum()))
