#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m41; mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "skinny or decode_gemm" 2>&1 | tail -3
for lib in product wd1 wd2 product wd1; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_$lib.log 2>&1
  echo "$lib: $(tail -1 $O/b_$lib.log | cut -c1-120)"
done
