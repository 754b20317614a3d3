#!/bin/bash
# Round-2 pass 3: attention v3 correctness + A/B, multi-region preprocessing debug, GEMM de-phase variants.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m3
mkdir -p $O
GAR_ATTN_V3=0 python tools/debug_multi_preproc.py > $O/debug_multi.log 2>&1
GAR_ATTN_V3=0 python tools/debug_batch_rows.py > $O/debug_rows.log 2>&1
GAR_ATTN_V3=0 python tools/debug_batch_rows.py full > $O/debug_rows_full.log 2>&1
( timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" --timeout=600 ) > $O/pytest_attn.log 2>&1
tail -5 $O/pytest_attn.log
echo "--- v3" > $O/attn.log; python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- v2" >> $O/attn.log; GAR_ATTN_V3=0 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- v3 again" >> $O/attn.log; python tools/bench_attn.py >> $O/attn.log 2>&1
cat $O/attn.log
for v in "" dephase1 dephase2; do
  echo "--- gemm variant: ${v:-product}" >> $O/gemm.log
  if [ -n "$v" ]; then export GAR_HIP_LIB=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants/libgar_hip_$v.so; else unset GAR_HIP_LIB; fi
  REPS=5 SHAPES=9 python tools/bench_gemm.py >> $O/gemm.log 2>&1
done
unset GAR_HIP_LIB
cat $O/gemm.log
( timeout 1800 python -m pytest tests/test_gpu_ops.py tests/test_gpu_e2e.py -q --timeout=900 ) > $O/pytest_ops_e2e.log 2>&1
tail -12 $O/pytest_ops_e2e.log
( GAR_ATTN_V3=0 timeout 1800 python -m pytest tests/test_gpu_e2e.py -q --timeout=900 ) > $O/pytest_e2e_v2.log 2>&1
tail -8 $O/pytest_e2e_v2.log
