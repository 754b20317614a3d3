#!/bin/bash
# Round-2 measurement pass 1 (run through gpurun): GPU tests, baseline micro-benchmarks, clock/power trace, PMC pass.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m1
mkdir -p $O
( time timeout 2400 python -m pytest tests -m gpu -q --timeout=1200 -s 2>&1 | tail -80 ) > $O/pytest.log 2>&1
python tools/bench_skinny.py 64 > $O/skinny.log 2>&1
python tools/bench_attn.py > $O/attn.log 2>&1
REPS=3 python tools/bench_gemm.py > $O/gemm.log 2>&1
python tools/smi_trace.py $O/smi_bench.json -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench.log 2>&1
python tools/smi_trace.py $O/smi_gemm.json -- env REPS=20 python tools/bench_gemm.py > $O/gemm_smi.log 2>&1
cd /tmp
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT \
   -d $GRAFT_REPO_ROOT/$O/pmcA --output-format csv -- env REPS=2 python $GRAFT_REPO_ROOT/tools/bench_gemm.py > $GRAFT_REPO_ROOT/$O/pmcA.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum \
   -d $GRAFT_REPO_ROOT/$O/pmcB --output-format csv -- env REPS=2 python $GRAFT_REPO_ROOT/tools/bench_gemm.py > $GRAFT_REPO_ROOT/$O/pmcB.log 2>&1
timeout 900 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE GRBM_COUNT \
   -d $GRAFT_REPO_ROOT/$O/pmcAttn --output-format csv -- python $GRAFT_REPO_ROOT/tools/bench_attn.py > $GRAFT_REPO_ROOT/$O/pmcAttn.log 2>&1
cd $GRAFT_REPO_ROOT
for d in pmcA pmcB pmcAttn; do
  c=$(find $O/$d -name '*counter_collection.csv' | head -1); k=$(find $O/$d -name '*kernel_trace.csv' | head -1)
  [ -n "$c" ] && python tools/pmc_kernels.py $c $k > $O/$d.json 2> $O/$d.err
  # keep the merged-back payload small: raw per-dispatch CSVs can be large
  find $O/$d -name '*.csv' -size +20M -delete
done
ls -la $O
