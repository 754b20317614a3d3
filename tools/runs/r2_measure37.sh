#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "qkv_rope or vit_attention" 2>&1 | tail -3
for lib in product divrow product divrow; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib"; SHAPESET=plan SHAPES=1 timeout 300 python tools/bench_gemm.py 2>&1 | grep -E "qkv"
done
