#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention or qkv_rope" 2>&1 | tail -3
for v in 0 1 0 1; do echo "== VROW=$v"; VROW=$v SHAPESET=all timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep 8b; done
for f in "" "--vit-v-transpose" ""; do
  timeout 900 python bench.py --no-cpu-baseline --model gar_8b --max-num-tiles 8 --steps 2 $f > gpurun_out/b8.log 2>&1
  echo "bench 8b [$f]: $(tail -1 gpurun_out/b8.log | cut -c1-110)"
done
