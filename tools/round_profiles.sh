#!/bin/bash
# Evidence for profiles/: kernel-trace stats of the headline command (GAR-1B) and of GAR-8B, the two PMC traffic passes,
# SQ/GRBM counter passes of the GEMM and attention micro-benchmarks, a clock/power trace, and the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=${R:-r6}
O=$GRAFT_REPO_ROOT/gpurun_out/prof
mkdir -p $O
python bench.py --steps 3 --warmup 1 > $O/bench_default.log 2>&1; tail -1 $O/bench_default.log | cut -c1-300
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt1b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $O/kt1b.log 2>&1
timeout 900 rocprofv3 --kernel-trace --stats -d $O/kt8b --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --model gar_8b --max-num-tiles 8 --steps 2 > $O/kt8b.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_fetch --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --new-tokens 1 --no-graph > $O/pmc_fetch.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_write --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 1 --warmup 0 --new-tokens 1 --no-graph > $O/pmc_write.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT \
   -d $O/pmc_sq_gemm --output-format csv -- env REPS=2 SHAPES=9 python $GRAFT_REPO_ROOT/tools/bench_gemm.py > $O/pmc_sq_gemm.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT \
   -d $O/pmc_sq_attn --output-format csv -- env VROW=1 KV_PREFIX=1 python $GRAFT_REPO_ROOT/tools/bench_attn.py > $O/pmc_sq_attn.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/smi_trace.py $O/smi_bench.json -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/smi_bench.log 2>&1
find $O/kt1b -name '*kernel_stats.csv' -exec cp {} $O/${R}_kernel_stats.csv \;
find $O/kt8b -name '*kernel_stats.csv' -exec cp {} $O/${R}_kernel_stats_gar8b.csv \;
f=$(find $O/pmc_fetch -name '*counter_collection.csv' | head -1); w=$(find $O/pmc_write -name '*counter_collection.csv' | head -1)
python tools/pmc_summary.py $f $w $O/${R}_pmc_traffic.json 2.0 > $O/pmc_summary.log 2>&1
for d in pmc_sq_gemm pmc_sq_attn; do
  c=$(find $O/$d -name '*counter_collection.csv' | head -1); k=$(find $O/$d -name '*kernel_trace.csv' | head -1)
  python tools/pmc_kernels.py $c $k > $O/${R}_$d.json 2> $O/$d.err
done
R=$R python - <<'PY'
import json,os
O=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/prof'
d=json.load(open(O+'/smi_bench.json')); json.dump(d['summary'], open(O+'/'+os.environ.get('R','r3')+'_smi_bench_summary.json','w'), indent=1)
PY
find $O -name '*.csv' -size +3M -delete; find $O -name '*.db' -delete; rm -f $O/smi_bench.json
ls -la $O; cat $O/pmc_summary.log
# bench lines of the other workloads (GAR-8B single, 4-mask relationship prompt, 8-frame video)
timeout 900 python bench.py --no-cpu-baseline --model gar_8b --max-num-tiles 8 --steps 2 > $O/bench_gar8b.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --workload multi_region --steps 2 > $O/bench_multi.log 2>&1
timeout 900 python bench.py --no-cpu-baseline --workload video --steps 2 > $O/bench_video.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --batch 1 --steps 5 --warmup 2 > $O/bench_batch1.log 2>&1
tail -1 $O/bench_batch1.log > $O/${R}_bench_batch1.json
timeout 600 python bench.py --eos-mix > $O/bench_eosmix.log 2>&1
tail -1 $O/bench_eosmix.log > $O/${R}_bench_eos_mix.json
tail -1 $O/bench_default.log > $O/${R}_bench_default.json
tail -1 $O/bench_gar8b.log > $O/${R}_bench_gar8b.json
tail -1 $O/bench_multi.log > $O/${R}_bench_multi_region.json
tail -1 $O/bench_video.log > $O/${R}_bench_video_gar8b.json
for f in $O/${R}_bench_*.json; do echo "$f: $(cut -c1-150 $f)"; done
