#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m34
mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; tail -3 $O/tests.log
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product exactmax product; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_$lib.log 2>&1
  echo "$lib: $(tail -1 $O/b_$lib.log | cut -c1-140)"
done
