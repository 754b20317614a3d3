#!/bin/bash
# Attention segment timeline (tools/attn_timeline.py) and the GEMM K-tile timeline on more shapes (is the slow start of an
# output tile the cold A panel?).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m12
mkdir -p $O
VD=$PWD/grasp-any-region_amd/gar_amd/variants
GAR_HIP_LIB=$VD/libgar_hip_attntl.so timeout 300 python tools/attn_timeline.py > $O/attn_timeline.txt 2>&1
grep -v amdgpu.ids $O/attn_timeline.txt
timeout 300 python tools/bench_attn.py 2>&1 | grep -v amdgpu.ids | tee $O/attn_product.txt
GAR_HIP_LIB=$VD/libgar_hip_tl4.so TILEPOS=1 MORE_SHAPES=1 timeout 300 python tools/gemm_timeline.py > $O/gemm_tl4.txt 2>&1
grep -v amdgpu.ids $O/gemm_tl4.txt
