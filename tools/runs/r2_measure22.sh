#!/bin/bash
# planner A/B: region chunks of 16 (old) vs round-aware passes (new), same box, alternating
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m27
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1
tail -3 $O/tests.log
for c in 16 0 16 0; do
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --prefill-chunk $c > $O/b_c$c.log 2>&1
  echo "chunk $c: $(tail -1 $O/b_c$c.log | cut -c1-140)"
done
tail -1 $O/b_c0.log > $O/bench_planner.json
