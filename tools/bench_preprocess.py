#!/usr/bin/env python
"""Per-region cost of building one model input sample (1024^2 image + mask -> 17 tiles x 2, prompt ids, bbox):
host preprocessing (torch CPU bicubic) vs device preprocessing (preprocess.hip), and the device kernels alone."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "grasp-any-region_amd"))
import torch  # noqa: E402

from gar_amd import GARConfig, hip  # noqa: E402
from gar_amd.eval_dataset import SingleRegionCaptionDataset  # noqa: E402
from gar_amd.processing import GARProcessor  # noqa: E402
from gar_amd.synthetic import synthetic_image, synthetic_mask  # noqa: E402


def main():
    hip.require_device(0)
    cfg = GARConfig.gar_1b()
    n = 8
    imgs = [(synthetic_image(i), synthetic_mask(i)) for i in range(n)]
    ph = GARProcessor.from_config(cfg, 16)
    pg = GARProcessor.from_config(cfg, 16).use_gpu_preprocessing("cuda:0", torch.bfloat16)
    for name, proc, dev in (("host", ph, "cpu"), ("device", pg, "cuda:0")):
        SingleRegionCaptionDataset(*imgs[0], proc, data_dtype=torch.bfloat16, device=dev)[0]      # warm tables
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for im, m in imgs:
            s = SingleRegionCaptionDataset(im, m, proc, data_dtype=torch.bfloat16, device=dev)[0]
            if dev == "cpu":
                s = {k: (v.to("cuda:0") if torch.is_tensor(v) else v) for k, v in s.items()}      # what the model needs
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        print(f"{name:7s} sample build (incl. upload): {dt * 1e3:8.1f} ms/region  -> {1 / dt:6.1f} regions/s per process")
    ip = pg.image_processor
    im = imgs[0][0]
    ip(im, "bicubic")
    src = ip._upload(im)
    out = torch.empty(17, 3, 448, 448, dtype=torch.bfloat16, device="cuda:0")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ip._resize_into(src, out, 1, 1, 0, "bicubic")
        ip._resize_into(src, out, 4, 4, 1, "bicubic")
        ip._resize_into(src, out, 1, 1, 0, "nearest")
        ip._resize_into(src, out, 4, 4, 1, "nearest")
    e1.record()
    torch.cuda.synchronize()
    print(f"device kernels alone (image bicubic + id-matrix nearest, 2 x 17 tiles): {e0.elapsed_time(e1) / 20:.3f} ms/region")


if __name__ == "__main__":
    main()
