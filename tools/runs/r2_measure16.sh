#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m17
mkdir -p $O
( timeout 2400 python -m pytest tests -q -x -m gpu --timeout=900 ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-200
