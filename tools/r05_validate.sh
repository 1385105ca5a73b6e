#!/bin/bash
# round 5: validation of a kernel change on one box: the GPU suite under both contraction contracts, a soak of the rolling launch,
# random stress against the oracle
cd $GRAFT_REPO_ROOT
tag=${1:-r05v}
( time python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests.log 2>&1; tail -n 5 gpurun_out/${tag}_gputests.log | head -2
( time FASTKV_CONTRACTION=fmaf python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests_fmaf.log 2>&1; tail -n 5 gpurun_out/${tag}_gputests_fmaf.log | head -2
python tools/soak_rolling.py 300 71 2>&1 | tail -n 1 | tee gpurun_out/${tag}_soak.log
python tools/stress_parity.py 1500 72 2>&1 | grep "cases," | tee gpurun_out/${tag}_stress.log
