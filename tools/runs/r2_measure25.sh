#!/bin/bash
# gather patch-embed: tests, GEMM regression check (kernel text changed for every instantiation), bench A/B via w_patch_gather
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m30
mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "patch_embed or gemm" > $O/tests_ops.log 2>&1; tail -5 $O/tests_ops.log
SHAPES=8 timeout 300 python tools/bench_gemm.py 2>&1 | grep -v amdgpu | cut -c1-90
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/b_gather.log 2>&1
echo "gather: $(tail -1 $O/b_gather.log | cut -c1-140)"
timeout 600 python bench.py --no-cpu-baseline --no-patch-gather --steps 3 --warmup 1 > $O/b_im2col.log 2>&1
echo "im2col: $(tail -1 $O/b_im2col.log | cut -c1-140)"
