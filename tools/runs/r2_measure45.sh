#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product nt2a nt2b nt2c product; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib 8b"; MODEL=8b COLD=1 timeout 300 python tools/bench_skinny.py 64 2>&1 | grep -E "^M= 64" | head -4
  echo "== $lib 1b"; COLD=1 timeout 300 python tools/bench_skinny.py 64 2>&1 | grep -E "^M= 64" | head -4
done
