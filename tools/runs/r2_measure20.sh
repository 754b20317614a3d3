#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m25
mkdir -p $O
timeout 900 python bench.py --no-cpu-baseline --steps 20 --warmup 2 > $O/soak.log 2>&1; tail -1 $O/soak.log | cut -c1-170
timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --preprocess device > $O/dev.log 2>&1; tail -1 $O/dev.log | cut -c1-170
timeout 900 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/res.log 2>&1; tail -1 $O/res.log | cut -c1-170
python - <<'PY'
import sys, os, torch
sys.path.insert(0, "grasp-any-region_amd")
from gar_amd import GARConfig
from gar_amd.modeling_gar import GARModel
m = GARModel.from_shapes(GARConfig.gar_1b(), torch.bfloat16)
print("weights + tables resident:", round(torch.cuda.memory_allocated() / 2**30, 2), "GiB")
PY
