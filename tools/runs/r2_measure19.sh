#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m23
mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "gemm or norm" --timeout=600 ) 2>&1 | tail -3
timeout 300 python tools/bench_skinny.py 64 2>&1 | grep -v amdgpu.ids | tee $O/skinny.txt
( timeout 1800 python -m pytest tests/test_gpu_e2e.py -q -x --timeout=900 ) 2>&1 | tail -3
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-160
