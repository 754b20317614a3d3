#!/bin/bash
# Final evidence of the round: everything tools/r2_profiles.sh collects + the bench lines of the other workloads.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/r2_profiles.sh > gpurun_out/prof_run.log 2>&1
O=$GRAFT_REPO_ROOT/gpurun_out/prof
cd "$GRAFT_REPO_ROOT"
timeout 900 python bench.py --no-cpu-baseline --model gar_8b --max-num-tiles 8 --steps 2 > $O/bench_gar8b.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --workload multi_region --steps 2 > $O/bench_multi.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --workload video --steps 2 > $O/bench_video.log 2>&1
tail -1 $O/bench_default.log > $O/r2_bench_default.json
tail -1 $O/bench_gar8b.log > $O/r2_bench_gar8b.json
tail -1 $O/bench_multi.log > $O/r2_bench_multi_region.json
tail -1 $O/bench_video.log > $O/r2_bench_video_gar8b.json
for f in $O/r2_bench_*.json; do echo "$f: $(cut -c1-150 $f)"; done
tail -5 gpurun_out/prof_run.log
