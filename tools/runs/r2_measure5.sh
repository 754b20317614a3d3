#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m5
mkdir -p $O
GAR_ATTN_V3=0 python tools/debug_vit_rows.py > $O/debug_vit_rows.log 2>&1
grep -v amdgpu.ids $O/debug_vit_rows.log | grep -v SAME
( GAR_ATTN_V3=0 timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x --timeout=600 ) > $O/pytest_ops.log 2>&1
tail -5 $O/pytest_ops.log
REPS=5 SHAPES=9 GAR_ATTN_V3=0 python tools/bench_gemm.py > $O/gemm.log 2>&1
grep -v amdgpu.ids $O/gemm.log
GAR_ATTN_V3=0 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.log 2>&1
tail -1 $O/bench.log | cut -c1-300
( GAR_ATTN_V3=0 timeout 1800 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_demo.py tests/test_gpu_bench_loops.py tests/test_gpu_preprocess.py -q --timeout=900 ) > $O/pytest_rest.log 2>&1
tail -8 $O/pytest_rest.log
