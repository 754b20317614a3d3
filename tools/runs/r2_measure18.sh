#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m22
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "llm or decode or attention" --timeout=600 ) 2>&1 | tail -2
( timeout 1800 python -m pytest tests/test_gpu_e2e.py -q -x --timeout=900 ) 2>&1 | tail -2
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-160
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench2.log 2>&1; tail -1 $O/bench2.log | cut -c1-160
