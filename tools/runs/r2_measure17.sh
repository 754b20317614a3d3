#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m19
mkdir -p $O
VD=$PWD/grasp-any-region_amd/gar_amd/variants
( timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "norm" --timeout=600 ) 2>&1 | tail -2
for r in 1 2; do
echo "--- new (R rows per wave, raw-cached)"; timeout 300 python tools/bench_norm.py 2>&1 | grep -v amdgpu.ids
echo "--- old"; GAR_HIP_LIB=$VD/libgar_hip_normold.so timeout 300 python tools/bench_norm.py 2>&1 | grep -v amdgpu.ids
done
