#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m8
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 -s ) > $O/pytest.log 2>&1
tail -12 $O/pytest.log
python tools/debug_batch_rows.py full 2>&1 | grep -v amdgpu.ids | cut -c1-300 > $O/debug_rows_full.log; cat $O/debug_rows_full.log
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_single.log 2>&1; tail -1 $O/bench_single.log | cut -c1-200
python bench.py --no-cpu-baseline --model gar_8b --max-num-tiles 8 --steps 2 --warmup 1 > $O/bench_8b.log 2>&1; tail -1 $O/bench_8b.log | cut -c1-200
python bench.py --no-cpu-baseline --workload video --batch 16 --steps 2 --warmup 1 > $O/bench_video.log 2>&1; tail -1 $O/bench_video.log | cut -c1-200
python tools/bench_attn.py > $O/attn.log 2>&1; grep -v amdgpu.ids $O/attn.log
