#!/bin/bash
# A/B of the attention micro-optimisation (permlane32_swap half exchange + split max/sum chains) against the
# previous kernel (variants/libgar_hip_attnprev.so), then the full GPU suite and the default bench line.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/m10
mkdir -p $O
V=$PWD/grasp-any-region_amd/gar_amd/variants/libgar_hip_attnprev.so
( timeout 1200 python -m pytest tests/test_gpu_ops.py -q -x -k "attention or attn" --timeout=600 ) > $O/pytest_attn.log 2>&1
tail -3 $O/pytest_attn.log
echo "--- new" > $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- prev" >> $O/attn.log; GAR_HIP_LIB=$V timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- new again" >> $O/attn.log; timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
echo "--- prev again" >> $O/attn.log; GAR_HIP_LIB=$V timeout 300 python tools/bench_attn.py >> $O/attn.log 2>&1
grep -v amdgpu.ids $O/attn.log
timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_new.log 2>&1; tail -1 $O/bench_new.log | cut -c1-160
GAR_HIP_LIB=$V timeout 600 python bench.py --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_prev.log 2>&1; tail -1 $O/bench_prev.log | cut -c1-160
( timeout 2400 python -m pytest tests -q -x -m gpu --timeout=900 ) > $O/pytest_gpu.log 2>&1
tail -5 $O/pytest_gpu.log
