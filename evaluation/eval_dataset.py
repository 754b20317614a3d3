"""Drop-in import path of the reference's sample builders (``from evaluation.eval_dataset import ...`` as in
demo/gar_with_mask.py:15 and evaluation/*/inference.py of the reference); the implementation lives in
gar_amd.eval_dataset."""
import os
import sys

_PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "grasp-any-region_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from gar_amd.eval_dataset import (MultiRegionDataset, SingleRegionCaptionDataset,  # noqa: E402,F401
                                  VideoRegionCaptionDataset)
