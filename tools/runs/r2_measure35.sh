#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out/m37
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/m37/tests.log 2>&1; tail -3 gpurun_out/m37/tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash tools/r2_final.sh > gpurun_out/m37/final.log 2>&1
tail -3 gpurun_out/m37/final.log
python bench.py > gpurun_out/bench_final.log 2>&1; tail -1 gpurun_out/bench_final.log > gpurun_out/r2_bench_default.json; cut -c1-200 gpurun_out/r2_bench_default.json
