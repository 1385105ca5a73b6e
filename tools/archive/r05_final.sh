#!/bin/bash
# end-of-round validation on one box: the whole GPU suite, the profile round (bench, rocprofv3 trace, PMC passes), smoke
cd $GRAFT_REPO_ROOT
tag=${1:-r05z}
( time python -m pytest tests -q -m gpu ) > gpurun_out/${tag}_gputests.log 2>&1
tail -n 6 gpurun_out/${tag}_gputests.log
bash tools/profile_round.sh $tag > gpurun_out/${tag}_profile_round.log 2>&1
tail -n 1 gpurun_out/${tag}_profile_round.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 1
