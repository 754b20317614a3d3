import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "grasp-any-region_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


def _hf_style_checkpoint(cfg, W, vision_bias="split"):
    """The tensors and config.json a released checkpoint would hold: vision model_args with only the overrides the
    reference reads (depth / mlp_dim left to the timm architecture), the fused-qkv bias in timm Eva's q_bias/v_bias
    form (or un-fused q/k/v projections), plus tensors the path never reads."""
    D = cfg.mllm_config.vision_config.embed_dim
    import torch
    out = {}
    for k, t in W.items():
        if k.endswith("attn.qkv.bias") and vision_bias == "split":
            a = k[:-len("qkv.bias")]
            out[a + "q_bias"], out[a + "k_bias"], out[a + "v_bias"] = t[:D].clone(), t[D:2 * D].clone(), t[2 * D:].clone()
        elif k.endswith("attn.qkv.bias") and vision_bias == "unfused":
            a = k[:-len("qkv.bias")]
            for i, n in enumerate("qkv"):
                out[a + f"{n}_proj.bias"] = t[i * D:(i + 1) * D].clone()
        elif k.endswith("attn.qkv.weight") and vision_bias == "unfused":
            a = k[:-len("qkv.weight")]
            for i, n in enumerate("qkv"):
                out[a + f"{n}_proj.weight"] = t[i * D:(i + 1) * D].clone()
        else:
            out[k] = t
    out["mllm.model.vision_tower.timm_model.rope.periods"] = torch.zeros(8)         # ignored extras
    out["mllm.model.language_model.rotary_emb.inv_freq"] = torch.zeros(32)
    d = cfg.to_dict()
    margs = d["mllm_config"]["vision_config"]["model_args"]
    for k in ("depth", "mlp_dim"):
        margs.pop(k)
    d["mllm_config"]["vision_config"]["model_type"] = "timm_wrapper"
    d["architectures"] = ["GARModel"]
    return out, d


@pytest.fixture(scope="session")
def hf_style_checkpoint():
    return _hf_style_checkpoint
