#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m48; mkdir -p $O
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" 2>&1 | tail -3
for rep in 1 2; do
for lib in product gm8; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  echo "== $lib $(SHAPESET=plan timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids | awk '{printf "%s_%s=%s ", $1,$2,$(NF-3)}' | cut -c1-300)"
done; done
for lib in product gm8 product gm8; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_$lib.log 2>&1
  echo "$lib: $(tail -1 $O/b_$lib.log | cut -c1-120)"
done
