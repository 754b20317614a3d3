#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m9
mkdir -p $O
( timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" --timeout=600 ) > $O/pytest_attn.log 2>&1
tail -5 $O/pytest_attn.log
echo "--- pp" > $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- v2" >> $O/attn.log; GAR_ATTN_PP=0 timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- pp again" >> $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
grep -v amdgpu.ids $O/attn.log
( timeout 1800 python -m pytest tests/test_gpu_e2e.py -q -x --timeout=900 ) > $O/pytest_e2e.log 2>&1
tail -5 $O/pytest_e2e.log
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_pp.log 2>&1; tail -1 $O/bench_pp.log | cut -c1-200
GAR_ATTN_PP=0 timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_v2.log 2>&1; tail -1 $O/bench_v2.log | cut -c1-200
timeout 600 python bench.py --no-cpu-baseline --steps 2 --warmup 1 --batch 128 > $O/bench_b128.log 2>&1; tail -1 $O/bench_b128.log | cut -c1-200
