#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product qbf product qbf; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib $(SHAPESET=plan SHAPES=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -E "qkv" | cut -c1-75 | tr '\n' '|')"
done
export GAR_HIP_LIB=$V/libgar_hip_qbf.so
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "qkv_rope" 2>&1 | tail -3
