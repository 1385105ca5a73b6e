#!/bin/bash
# round 6, fourth contact: (1) what the explicit hand-over of the record areas is for -- the slow-entry test on an experiments build with the
# hand-over switched off (expected: the launch gives up its waits and is REPORTED); (2) the library with the four-operation fixed-point
# conversion: timing; (3) the whole GPU suite under the new default contract (fmaf) and under FASTKV_CONTRACTION=mfma16
cd $GRAFT_REPO_ROOT
( FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_exp FASTKV_FUSED_NO_HANDOVER=1 FASTKV_SPIN_LIMIT_MS=300 timeout 600 python -m pytest tests/test_rolling_gpu.py -q -m gpu -k "slow_entry" 2>&1 | grep -v "^$" | tail -n 25 ) > gpurun_out/r06d_no_handover.log 2>&1
tail -n 6 gpurun_out/r06d_no_handover.log
for r in 0 1; do FASTKV_FUSED_ROLLING=$r python tools/exp_interleave.py 2>&1 | grep "B=8\|B=16"; done | tee gpurun_out/r06d_interleave.log
( time python -m pytest tests -q -m gpu ) > gpurun_out/r06d_gputests.log 2>&1; tail -n 8 gpurun_out/r06d_gputests.log
( time FASTKV_CONTRACTION=mfma16 python -m pytest tests -q -m gpu ) > gpurun_out/r06d_gputests_mfma16.log 2>&1; tail -n 8 gpurun_out/r06d_gputests_mfma16.log
