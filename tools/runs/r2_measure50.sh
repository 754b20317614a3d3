#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product resf product resf; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib $(SHAPESET=plan timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -E "proj|fc2|llm o|down" | awk '{printf "%s_%s=%s ", $1,$2,$(NF-3)}')"
done
export GAR_HIP_LIB=$V/libgar_hip_resf.so
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
