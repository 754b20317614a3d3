#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m14
mkdir -p $O
VD=$PWD/grasp-any-region_amd/gar_amd/variants
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" --timeout=600 ) > $O/pytest_attn.log 2>&1
tail -2 $O/pytest_attn.log
echo "--- product (one workgroup per item)" > $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
for n in 2 3 4 5; do echo "--- persistent, $n workgroups per CU" >> $O/attn.log; GAR_HIP_LIB=$VD/libgar_hip_attnp$n.so timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1; done
echo "--- product again" >> $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
grep -v amdgpu.ids $O/attn.log
for n in 3 4; do ( GAR_HIP_LIB=$VD/libgar_hip_attnp$n.so timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" --timeout=600 ) 2>&1 | tail -1; done
