#!/bin/bash
# round 5, GPU run 1: the new tests first, then the whole GPU suite, two consecutive bench lines (is the CPU leg reproducible?), the 8-rank rehearsal
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
( time python -m pytest tests/test_model_parity_gpu.py tests/test_rolling_gpu.py -x -q -m gpu -k "published or abandoned" ) > gpurun_out/r05a_newtests.log 2>&1
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r05a_gputests.log 2>&1
( time python bench.py ) > gpurun_out/r05a_bench.json 2> gpurun_out/r05a_bench.err
( time python bench.py --no-ttft ) > gpurun_out/r05a_bench_2.json 2> gpurun_out/r05a_bench_2.err
( time BENCH_BACKEND=gloo python bench.py --gpus 8 ) > gpurun_out/r05a_bench8_gloo.json 2> gpurun_out/r05a_bench8_gloo.err
tail -3 gpurun_out/r05a_newtests.log gpurun_out/r05a_gputests.log; cut -c1-300 gpurun_out/r05a_bench.json; tail -4 gpurun_out/r05a_bench8_gloo.err
