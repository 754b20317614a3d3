#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m36
mkdir -p $O
for v in 0 1 0 1; do echo "== VROW=$v"; VROW=$v timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids; done
for f in "" "--vit-v-transpose" "" "--vit-v-transpose"; do
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 $f > $O/b.log 2>&1
  echo "bench [$f]: $(tail -1 $O/b.log | cut -c1-120)"
done
