#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product gm4 gm16 gm2 product; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib"; SHAPESET=plan timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | cut -c1-85
done
