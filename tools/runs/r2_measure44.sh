#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product wd1 wd2 product wd1; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib"; MODEL=8b COLD=1 timeout 300 python tools/bench_skinny.py 64 2>&1 | grep -E "^M= 64" | head -6
done
