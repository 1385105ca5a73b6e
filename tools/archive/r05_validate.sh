#!/bin/bash
# round 5: validation of a kernel change on one box: the GPU suite (under both contraction contracts unless SHORT=1), a soak of the rolling
# launch, random stress against the oracle
cd $GRAFT_REPO_ROOT
tag=${1:-r05v}
( time python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests.log 2>&1; tail -n 5 gpurun_out/${tag}_gputests.log | head -2
if [ -z "$SHORT" ]; then ( time FASTKV_CONTRACTION=fmaf python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests_fmaf.log 2>&1; tail -n 5 gpurun_out/${tag}_gputests_fmaf.log | head -2; fi
python tools/soak_rolling.py ${SOAK_S:-300} 71 2>&1 | tail -n 1 | tee gpurun_out/${tag}_soak.log
python tools/stress_parity.py ${STRESS_N:-1500} 72 2>&1 | grep "cases," | tee gpurun_out/${tag}_stress.log
