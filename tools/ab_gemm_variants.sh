#!/bin/bash
# usage: abrun.sh REPS ROUNDS variant...
V=grasp-any-region_amd/gar_amd/variants
reps=$1; rounds=$2; shift 2
for r in $(seq $rounds); do for v in "$@"; do echo == $v; if [ $v = product ]; then REPS=$reps SHAPESET=plan python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids; else REPS=$reps GAR_HIP_LIB=$V/libgar_hip_$v.so SHAPESET=plan python tools/bench_gemm.py 2>&1 | grep -v amdgpu.ids; fi; done; done
