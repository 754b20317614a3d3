#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m26
mkdir -p $O
for cfg in "64 16" "60 12" "60 15" "63 9" "64 16" "60 12"; do
  set -- $cfg
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --batch $1 --prefill-chunk $2 > $O/b$1_c$2.log 2>&1
  echo "batch $1 chunk $2: $(tail -1 $O/b$1_c$2.log | cut -c1-125)"
done
