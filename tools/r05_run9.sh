#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r05i_gputests.log 2>&1
tail -n 6 gpurun_out/r05i_gputests.log
bash tools/profile_round.sh r05i > gpurun_out/r05i_profile_round.log 2>&1
tail -n 2 gpurun_out/r05i_profile_round.log | cut -c1-300
