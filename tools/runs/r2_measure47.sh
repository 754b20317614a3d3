#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for rep in 1 2; do
for lib in product gm1 gm2 gm4 gm16 gm32; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib $(SHAPESET=plan timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | grep -v weighted | awk '{printf "%s_%s=%s ", $1,$2,$(NF-3)}')"
done; done
