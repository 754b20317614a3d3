"""debug: a batch of two IDENTICAL samples must give bit-identical rows — where does the first difference appear?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from gar_amd import GARConfig
from gar_amd.modeling_gar import GARModel
from gar_amd.processing import GARProcessor
from test_gpu_e2e import _sample
full = len(sys.argv) > 1 and sys.argv[1] == "full"
cfg = GARConfig.gar_1b() if full else GARConfig.gar_1b(**{"vision.depth": 2, "text.num_hidden_layers": 2})
proc = GARProcessor.from_config(cfg, max_num_tiles=16)
s = _sample(cfg, proc, 0, 1024, 1024, dtype=torch.bfloat16)
m = GARModel.from_synthetic(cfg, 0, torch.bfloat16)
two = dict(input_ids=torch.cat([s["input_ids"]] * 2), pixel_values=torch.cat([s["pixel_values"]] * 2),
           global_mask_values=torch.cat([s["global_mask_values"]] * 2), bboxes=s["bboxes"] * 2,
           aspect_ratios=torch.cat([s["aspect_ratios"]] * 2))
pv = two["pixel_values"].reshape(2, 17, 3, 448, 448)
gm = two["global_mask_values"].reshape(2, 17, 3, 448, 448)
proj = m.get_image_features(pv, gm, pooled=False).clone()
n = proj.shape[0] // 2
print("projector rows equal:", torch.equal(proj[:n], proj[n:]), float((proj[:n].float() - proj[n:].float()).abs().max()))
feats = m.get_image_features(pv, gm).clone()
print("pooled feats equal:", torch.equal(feats[:17], feats[17:]))
emb = m.build_inputs_embeds(two["input_ids"], None, two["bboxes"], two["aspect_ratios"], 17, True, None, proj=m.get_image_features(pv, gm, pooled=False)).clone()
print("embeds equal:", torch.equal(emb[0], emb[1]), float((emb[0].float() - emb[1].float()).abs().max()))
for use_graph in (False, True):
    o = m.generate(**two, max_new_tokens=8, return_logits=True, use_graph=use_graph)
    d = (o.logits[0] - o.logits[1]).abs().amax(-1)
    print("graph" if use_graph else "eager", "per-step max |logit row0 - row1|:", [float(x) for x in d], o.sequences.tolist())
o1 = m.generate(**s, max_new_tokens=8, return_logits=True)
print("single vs batch row0 first-logit diff:", float((o1.logits[0, 0] - o.logits[0, 0]).abs().max()), o1.sequences.tolist())
o2 = m.generate(**s, max_new_tokens=8, return_logits=True)
print("single repeated: identical logits:", torch.equal(o1.logits, o2.logits))
