#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m13
mkdir -p $O
VD=$PWD/grasp-any-region_amd/gar_amd/variants
GAR_HIP_LIB=$VD/libgar_hip_attntl.so timeout 300 python tools/attn_timeline.py > $O/attn_timeline.txt 2>&1
grep -v amdgpu.ids $O/attn_timeline.txt
for c in 16 8 4 16 8; do
  timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 --prefill-chunk $c > $O/bench_c$c.log 2>&1; echo "chunk $c: $(tail -1 $O/bench_c$c.log | cut -c1-130)"
done
