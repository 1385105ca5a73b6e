#!/bin/bash
# round 5: A/B of the vector-instruction trims of score_fused, alternating on one box, eight 32k layers per call:
#   old = the library before them (build_x_old/), mid = v_fract in exp_to_fix2 + integer max pooling (build_x_mid/), new = mid + the
#   mixed-precision fma (v_fma_mix_f32) in the epilogue, phase B and phase C (in-tree)
cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  echo -n "old: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_old EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
  echo -n "mid: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_mid EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
  echo -n "new: "; EXP_B=8 python tools/exp_occ3.py 2>&1 | grep "us per call"
done 2>&1 | tee gpurun_out/r05_valu_ab.log
for sb in "2048 16" "32768 1"; do set -- $sb
  for l in old mid; do echo -n "$l S=$1 B=$2: "; FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_$l EXP_S=$1 EXP_B=$2 python tools/exp_occ3.py 2>&1 | grep "us per call"; done
  echo -n "new S=$1 B=$2: "; EXP_S=$1 EXP_B=$2 python tools/exp_occ3.py 2>&1 | grep "us per call"
done 2>&1 | tee -a gpurun_out/r05_valu_ab.log
