#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for o in 0 1 2 3 5; do FASTKV_FUSED_OCC3=$o timeout 300 python tools/exp_occ3.py; done > gpurun_out/r05b_occ3.log 2>&1
( time python -m pytest tests/test_rolling_gpu.py tests/test_hip_parity.py -x -q -m gpu -k "abandoned_rolling or half_the_chip or wide_sweep" ) > gpurun_out/r05b_tests.log 2>&1
( time BENCH_BACKEND=gloo python bench.py --gpus 8 ) > gpurun_out/r05b_bench8_gloo.json 2> gpurun_out/r05b_bench8_gloo.err
grep -v amdgpu.ids gpurun_out/r05b_occ3.log; tail -n 5 gpurun_out/r05b_tests.log; cut -c1-400 gpurun_out/r05b_bench8_gloo.json; tail -n 4 gpurun_out/r05b_bench8_gloo.err
