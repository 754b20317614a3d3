"""GPU (-m gpu): the demo CLIs (H2 of SURVEY.md section 8a) run end to end as subprocesses with seeded tiny weights —
device preprocessing (default) and host preprocessing print the same caption, and that caption is the decoded token
sequence the CPU oracle produces for the same inputs."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args):
    # PYTHONHASHSEED: the multi-region prompt order is the iteration order of a Python set of "<PromptK>" strings, in
    # the reference (evaluation/eval_dataset.py:205-212) and here — two processes only agree under the same hash seed
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(ROOT, "grasp-any-region_amd")]),
               PYTHONHASHSEED="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "demo", script), "--synthetic_weights", "--model_name_or_path",
                        "tiny", "--data_type", "fp32", "--max_num_tiles", "4", "--max_new_tokens", "12", *args],
                       capture_output=True, text=True, timeout=600, env=env, cwd=os.path.join(ROOT, "demo"))
    assert r.returncode == 0, r.stderr[-2000:]
    return r.stdout.strip().splitlines()[-1] if r.stdout.strip() else ""


def test_demo_clis(tmp_path):
    from gar_amd import GARConfig
    from gar_amd.eval_dataset import SingleRegionCaptionDataset
    from gar_amd.processing import GARProcessor
    from gar_amd.synthetic import synthetic_disjoint_masks, synthetic_image, synthetic_mask
    from gar_amd.weights import synthetic_weights
    from oracle import gar_oracle as O
    img = synthetic_image(11, 300, 220)
    ip = str(tmp_path / "img.png")
    img.save(ip)
    mp = str(tmp_path / "mask.png")
    Image.fromarray(synthetic_mask(11, 300, 220).astype(np.uint8) * 255).save(mp)
    dev_out = _run("gar_with_mask.py", "--image_path", ip, "--mask_path", mp)
    host_out = _run("gar_with_mask.py", "--image_path", ip, "--mask_path", mp, "--host_preprocessing")
    assert dev_out == host_out
    # oracle on the same inputs
    cfg = GARConfig.tiny()
    proc = GARProcessor.from_config(cfg, max_num_tiles=4)
    mask = np.array(Image.open(mp).convert("L")).astype(bool)
    s = SingleRegionCaptionDataset(Image.open(ip), mask, proc, data_dtype=torch.float32, device="cpu")[0]
    eos = proc.tokenizer.eos_token_id
    seq, _ = O.gar_generate(synthetic_weights(cfg, 0), cfg, s["pixel_values"], s["global_mask_values"], s["aspect_ratios"],
                            s["bboxes"], s["input_ids"], None, max_new_tokens=12, return_logits=True)
    ids = seq[0].tolist()
    if eos in ids:
        ids = ids[:ids.index(eos) + 1]
    assert dev_out == proc.tokenizer.decode(ids, skip_special_tokens=True).strip()
    # multi-region CLI runs and both preprocessing paths agree
    mps = []
    for k, m in enumerate(synthetic_disjoint_masks(11, 3, 300, 220)):
        p = str(tmp_path / f"m{k}.png")
        Image.fromarray(m.astype(np.uint8) * 255).save(p)
        mps.append(p)
    q = "What is the relationship between <Prompt0>, <Prompt1> and <Prompt2>?"
    a = _run("gar_relationship.py", "--image_path", ip, "--mask_paths", *mps, "--question_str", q)
    b = _run("gar_relationship.py", "--image_path", ip, "--mask_paths", *mps, "--question_str", q, "--host_preprocessing")
    assert a == b


def test_demo_cli_in_fp16_and_bf16(tmp_path):
    """--data_type fp16 | bf16 of the reference CLI (demo/gar_with_mask.py:41-45) run end to end (fp16: the twin library);
    at tiny dimensions both print the caption the fp32 run prints."""
    from gar_amd.synthetic import synthetic_image, synthetic_mask
    ip, mp = str(tmp_path / "img.png"), str(tmp_path / "mask.png")
    synthetic_image(12, 280, 200).save(ip)
    Image.fromarray(synthetic_mask(12, 280, 200).astype(np.uint8) * 255).save(mp)
    ref = _run("gar_with_mask.py", "--image_path", ip, "--mask_path", mp)
    for dt in ("fp16", "bf16"):
        out = _run("gar_with_mask.py", "--image_path", ip, "--mask_path", mp, "--data_type", dt)
        assert out, dt
        if dt == "fp16":
            assert out == ref          # 3e-3 logit error against margins of the tiny model: the same greedy caption
