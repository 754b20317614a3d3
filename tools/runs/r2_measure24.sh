#!/bin/bash
# A-stationary tile order (product) vs the 8 x 4 XCD block order (variant oldorder), planner shapes and the 8-region shapes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m29
mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for rep in 1 2; do
for lib in product oldorder; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  SHAPESET=plan timeout 300 python tools/bench_gemm.py > $O/plan_${lib}_$rep.log 2>&1
  echo "== plan shapes, $lib ($rep)"; grep -v amdgpu.ids $O/plan_${lib}_$rep.log | cut -c1-90
  SHAPES=4 timeout 300 python tools/bench_gemm.py > $O/r8_${lib}_$rep.log 2>&1
  echo "== 8-region ViT shapes, $lib ($rep)"; grep -v amdgpu.ids $O/r8_${lib}_$rep.log | cut -c1-90
done; done
unset GAR_HIP_LIB
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
