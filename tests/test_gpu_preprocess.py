"""GPU (-m gpu): device-side preprocessing against the host processor — bit-exact tiles."""
import numpy as np
import pytest
import torch
from PIL import Image

pytestmark = pytest.mark.gpu


def _img(seed, w, h, mode="RGB"):
    rng = np.random.default_rng(seed)
    a = rng.integers(0, 256, (h, w, 4 if mode == "RGBA" else 3), dtype=np.uint8)
    # smooth-ish content as well as noise: a gradient block
    a[: h // 2, : w // 2, :3] = (np.linspace(0, 255, w // 2)[None, :, None]).astype(np.uint8)
    return Image.fromarray(a, mode)


@pytest.mark.parametrize("w,h,tiles", [(1024, 1024, 16), (640, 480, 16), (333, 517, 8), (1500, 700, 16), (200, 160, 4)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_bicubic_tiles_bit_exact(w, h, tiles, dtype):
    from gar_amd.preprocess_gpu import GpuImageProcessor
    from gar_amd.processing import GARImageProcessor
    ts = 448 if tiles > 4 else 112
    host = GARImageProcessor(ts, tiles)
    dev = GpuImageProcessor(ts, tiles, device="cuda:0", dtype=dtype)
    im = _img(w * 7 + h, w, h)
    ref, ar = host(im, "bicubic")
    out, ar2 = dev(im, "bicubic")
    assert ar == ar2 and tuple(out.shape) == tuple(ref.shape)
    assert torch.equal(out.cpu().float(), ref.to(dtype).float())


@pytest.mark.parametrize("w,h", [(1024, 1024), (770, 1024), (123, 457)])
def test_nearest_id_matrix_bit_exact(w, h):
    from gar_amd.preprocess_gpu import GpuImageProcessor
    from gar_amd.processing import GARImageProcessor
    host = GARImageProcessor(448, 16)
    dev = GpuImageProcessor(448, 16, device="cuda:0", dtype=torch.bfloat16)
    rng = np.random.default_rng(w + h)
    ids = rng.integers(0, 6, (h, w), dtype=np.uint8)
    vp = Image.fromarray(np.repeat(ids[:, :, None], 3, axis=2), "RGB")
    ref, _ = host(vp, "nearest")
    out, _ = dev(vp, "nearest")
    assert torch.equal(out.cpu().float(), ref.to(torch.bfloat16).float())


def test_dataset_sample_identical_with_gpu_preprocessing():
    """the whole sample builder (eval_dataset.SingleRegionCaptionDataset) on the device path == host path, and the
    single-frame (video) entry point as well; RGBA input is converted like the reference does (:82)."""
    from gar_amd import GARConfig
    from gar_amd.eval_dataset import SingleRegionCaptionDataset
    from gar_amd.processing import GARProcessor
    from gar_amd.synthetic import synthetic_mask
    cfg = GARConfig.gar_1b()
    im = _img(5, 900, 700, "RGBA")
    mask = synthetic_mask(5, 900, 700)
    a = SingleRegionCaptionDataset(im, mask, GARProcessor.from_config(cfg, 16), data_dtype=torch.bfloat16, device="cpu")[0]
    pg = GARProcessor.from_config(cfg, 16).use_gpu_preprocessing("cuda:0", torch.bfloat16)
    b = SingleRegionCaptionDataset(im, mask, pg, data_dtype=torch.bfloat16, device="cuda:0")[0]
    assert b["pixel_values"].is_cuda and b["global_mask_values"].is_cuda
    for k in ("pixel_values", "global_mask_values", "input_ids", "aspect_ratios"):
        assert torch.equal(a[k].float(), b[k].cpu().float()), k
    assert a["bboxes"] == b["bboxes"]
    one_h = GARProcessor.from_config(cfg, 16).image_processor.single_tile(im)
    one_d = pg.image_processor.single_tile(im)
    assert torch.equal(one_d.cpu().float(), one_h.to(torch.bfloat16).float())
