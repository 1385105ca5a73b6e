#!/bin/bash
# round 5: groups of MID-LENGTH layers (4k - 16k tokens: the 16 layers behind the TSP layer under the published recipe are 6553 tokens
# each) -- the rolling launch at entries of four tiles per wave (the default) against entries of two / one tiles per wave with fewer
# entries on the chip (FASTKV_FUSED_ROLLING_PERT / _F: measurement switches), against launches in step (FASTKV_FUSED_ROLLING=0)
cd $GRAFT_REPO_ROOT
run() { echo -n "$* :: "; env "$@" python tools/exp_occ3.py 2>&1 | grep -E "us per call|Error|error" | head -3; }
for sb in "6553 16" "8192 16" "8192 8" "4096 16" "12288 8" "16384 8"; do
  set -- $sb
  echo "== S=$1 B=$2"
  run EXP_S=$1 EXP_B=$2
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_ROLLING=0
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_ROLLING_PERT=2 FASTKV_FUSED_ROLLING_F=4
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_ROLLING_PERT=2 FASTKV_FUSED_ROLLING_F=3
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_ROLLING_PERT=2 FASTKV_FUSED_ROLLING_F=2
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_ROLLING_PERT=1 FASTKV_FUSED_ROLLING_F=2
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_STAGGER_US=8
  run EXP_S=$1 EXP_B=$2 FASTKV_FUSED_STAGGER_US=20
done 2>&1 | tee gpurun_out/r05_midlen.log
