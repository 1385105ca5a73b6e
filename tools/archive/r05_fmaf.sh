#!/bin/bash
# the accuracy-work contract (FASTKV_CONTRACTION=fmaf) as the process default: the whole GPU suite and the bench line under it
cd $GRAFT_REPO_ROOT
export FASTKV_CONTRACTION=fmaf
( time python -m pytest tests -q -m gpu ) > gpurun_out/r05z_gputests_fmaf.log 2>&1
tail -n 5 gpurun_out/r05z_gputests_fmaf.log
python bench.py --no-ttft > gpurun_out/r05z_bench_fmaf.json 2> gpurun_out/r05z_bench_fmaf.err
cut -c1-260 gpurun_out/r05z_bench_fmaf.json
