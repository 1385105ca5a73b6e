#!/bin/bash
# round 6, eighth contact: the library as it ships (pacing off, tsp_propagate): the GPU suite under both defaults, the profile round, smoke
cd $GRAFT_REPO_ROOT
tag=${1:-r06z}
( time python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests.log 2>&1; tail -n 4 gpurun_out/${tag}_gputests.log | head -2
( time FASTKV_CONTRACTION=mfma16 python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests_mfma16.log 2>&1; tail -n 4 gpurun_out/${tag}_gputests_mfma16.log | head -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 2 | tee gpurun_out/${tag}_smoke.log
bash tools/profile_round.sh ${tag} > gpurun_out/${tag}_profile_round.log 2>&1; tail -n 1 gpurun_out/${tag}_profile_round.log | cut -c1-300
FASTKV_CONTRACTION=mfma16 python bench.py --no-ttft > gpurun_out/${tag}_bench_mfma16.json 2> gpurun_out/${tag}_bench_mfma16.err; cut -c1-200 gpurun_out/${tag}_bench_mfma16.json
