#!/bin/bash
# cache-policy bits on the operand DMA loads: A nt / A sc1 / W nt / A nt+sc1 vs product, planner shapes, same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m31
mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product ant asc1 wnt antsc product; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  SHAPESET=plan timeout 300 python tools/bench_gemm.py > $O/plan_${lib}.log 2>&1
  echo "== plan shapes, $lib"; grep -v amdgpu.ids $O/plan_${lib}.log | cut -c1-90
done
