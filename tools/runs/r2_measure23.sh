#!/bin/bash
# channel-camping experiment: row pitch of A / C padded by 64 B .. 256 B vs dense power-of-two pitches
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m28
mkdir -p $O
for cfg in "0 0" "64 0" "0 64" "64 64" "128 128" "32 32" "0 0"; do
  set -- $cfg
  PAD_A=$1 PAD_C=$2 SHAPES=9 timeout 300 python tools/bench_gemm.py > $O/pad_$1_$2.log 2>&1
  echo "== PAD_A=$1 PAD_C=$2"; cat $O/pad_$1_$2.log | cut -c1-90
done
