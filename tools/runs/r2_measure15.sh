#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m15
mkdir -p $O
VD=$PWD/grasp-any-region_amd/gar_amd/variants
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" --timeout=600 ) > $O/pytest_attn.log 2>&1
tail -2 $O/pytest_attn.log
for r in 1 2; do
echo "--- new (-m in the accumulator init)" >> $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- prev" >> $O/attn.log; GAR_HIP_LIB=$VD/libgar_hip_attnprev.so timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
done
grep -v amdgpu.ids $O/attn.log
