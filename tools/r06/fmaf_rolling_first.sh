#!/bin/bash
# round 6, first contact: the rolling launch under the fp32-fma-chain contract (FASTKV_CONTRACTION=fmaf)
cd $GRAFT_REPO_ROOT
export FASTKV_CONTRACTION=fmaf
out=gpurun_out/r06a_fmaf_rolling.log
: > $out
for r in 0 1; do FASTKV_FUSED_ROLLING=$r python tools/exp_interleave.py 2>&1 | grep ROLLING >> $out; done
for st in 8 18 24 30; do echo "stagger $st us" >> $out; FASTKV_FUSED_STAGGER_US=$st python tools/exp_interleave.py 2>&1 | grep "B=8\|B=16" >> $out; done
cat $out
( time timeout 900 python -m pytest tests/test_rolling_gpu.py -q -m gpu -x ) 2>&1 | tail -n 6 | tee gpurun_out/r06a_rolling_tests_fmaf.log
timeout 300 python tools/soak_rolling.py 150 61 2>&1 | tail -n 3 | tee gpurun_out/r06a_soak_fmaf.log
python bench.py --no-ttft > gpurun_out/r06a_bench_fmaf.json 2> gpurun_out/r06a_bench_fmaf.err
cut -c1-400 gpurun_out/r06a_bench_fmaf.json
