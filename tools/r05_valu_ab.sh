#!/bin/bash
# round 5: A/B of the vector-instruction trims (v_fract in exp_to_fix2, integer max pooling in phase D): the round's last library in
# build_x_old/ against the in-tree one, alternating, eight 32k layers per call; then the parity tests that cover both changes
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  echo -n "old: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_old EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
  echo -n "new: "; EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
done 2>&1 | tee gpurun_out/r05_valu_ab.log
for sb in "2048 16" "6553 16" "32768 1"; do set -- $sb
  echo -n "old S=$1 B=$2: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_old EXP_S=$1 EXP_B=$2 python tools/exp_occ3.py 2>&1 | grep "us per call"
  echo -n "new S=$1 B=$2: "; EXP_S=$1 EXP_B=$2 python tools/exp_occ3.py 2>&1 | grep "us per call"
done 2>&1 | tee -a gpurun_out/r05_valu_ab.log
( time python -m pytest tests/test_hip_parity.py tests/test_rolling_gpu.py tests/test_stress_gpu.py -q -m gpu -x ) 2>&1 | tail -5 | tee -a gpurun_out/r05_valu_ab.log
