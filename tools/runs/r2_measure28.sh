#!/bin/bash
# attention: lazy running max (product) vs the exact max on every tile (variant exactmax), same box, alternating; then the tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m33
mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product exactmax product exactmax; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib"; SHAPESET=all timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee -a $O/attn_$lib.log
done
unset GAR_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "attention" 2>&1 | tail -3
