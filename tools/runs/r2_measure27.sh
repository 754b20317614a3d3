#!/bin/bash
# epilogue: residual / pos-embed loads 2 (product) / 3 / 4 / 7 sixteen-row steps ahead of their use; planner shapes; same box
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m32
mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product aux3 aux4 aux7 product aux4; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  SHAPESET=plan timeout 300 python tools/bench_gemm.py > $O/plan_${lib}.log 2>&1
  echo "== plan shapes, $lib"; grep -v amdgpu.ids $O/plan_${lib}.log | grep -E "proj|fc2|llm o|down|weighted" | cut -c1-90
done
