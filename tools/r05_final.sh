#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests -q -m gpu ) > gpurun_out/r05z_gputests.log 2>&1
tail -n 6 gpurun_out/r05z_gputests.log
bash tools/profile_round.sh r05z > gpurun_out/r05z_profile_round.log 2>&1
tail -n 1 gpurun_out/r05z_profile_round.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
