#!/bin/bash
# Round-2 measurement pass 2: full GPU test log, the three bench workloads, kernel-trace stats of the headline command.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m2
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 -s ) > $O/pytest.log 2>&1
tail -15 $O/pytest.log
python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_single.log 2>&1
python bench.py --no-cpu-baseline --workload multi_region --steps 2 --warmup 1 > $O/bench_multi.log 2>&1
python bench.py --no-cpu-baseline --workload video --batch 16 --steps 2 --warmup 1 > $O/bench_video.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 2 --warmup 1 > $GRAFT_REPO_ROOT/$O/prof.log 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof -name '*kernel_stats.csv' -exec cp {} $O/kernel_stats.csv \;
find $O/prof -name '*.csv' -size +5M -delete
tail -2 $O/bench_single.log | cut -c1-400; tail -2 $O/bench_multi.log | cut -c1-400; tail -2 $O/bench_video.log | cut -c1-400
