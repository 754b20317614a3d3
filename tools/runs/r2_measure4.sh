#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m4
mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
GAR_ATTN_V3=0 python tools/debug_vit_rows.py > $O/debug_vit_rows.log 2>&1
cat $O/debug_vit_rows.log | grep -v amdgpu.ids
for v in tl4 tl4so; do echo "=== $v" >> $O/timeline.log; TILEPOS=1 GAR_HIP_LIB=$V/libgar_hip_$v.so python tools/gemm_timeline.py >> $O/timeline.log 2>&1; done
cat $O/timeline.log | grep -v amdgpu.ids
for v in "" storeov nostore l2store; do
  echo "--- gemm variant: ${v:-product}" >> $O/gemm.log
  if [ -n "$v" ]; then export GAR_HIP_LIB=$V/libgar_hip_$v.so; else unset GAR_HIP_LIB; fi
  REPS=5 python tools/bench_gemm.py >> $O/gemm.log 2>&1
done
unset GAR_HIP_LIB
cat $O/gemm.log | grep -v amdgpu.ids
