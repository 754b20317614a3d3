#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/grasp-any-region_amd/gar_amd/variants
for lib in product gm8 product gm8; do
  if [ $lib = product ]; then unset GAR_HIP_LIB; else export GAR_HIP_LIB=$V/libgar_hip_$lib.so; fi
  timeout 900 python bench.py --no-cpu-baseline --model gar_8b --max-num-tiles 8 --steps 2 > gpurun_out/b8.log 2>&1
  echo "8b $lib: $(tail -1 gpurun_out/b8.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],2), round(d['roofline']['achieved']))")"
done
