#!/bin/bash
# round 5: A/B of the vector-instruction trims of score_fused, alternating on one box, eight 32k layers per call:
#   old = the library before them (build_x_old/), mid = v_fract in exp_to_fix2 + integer max pooling (build_x_mid/), new = the in-tree
#   library (mid + the mixed-precision fma in the epilogue, phase B and phase C + the two-operation quotient of the epilogue)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "old: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_old EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
  echo -n "mid: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_mid EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
  echo -n "new: "; EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
done 2>&1 | tee gpurun_out/r05_valu_ab2.log
( time python -m pytest tests/test_hip_parity.py tests/test_rolling_gpu.py -q -m gpu -x ) 2>&1 | tail -5 | tee -a gpurun_out/r05_valu_ab2.log
