#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product fwdorder product fwdorder; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib"; VROW=1 SHAPESET=all timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | grep -E "vit"
done
unset GAR_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -3
