#!/bin/bash
cd $GRAFT_REPO_ROOT
export EXP_B=16 FASTKV_FUSED_CONVEYOR=1
for t in 0 2 4 6 8 16 32 56 62; do FASTKV_FUSED_TUNE=$t timeout 120 python tools/exp_occ3.py 2>&1 | grep -v amdgpu | tail -1; done
for d in 2 8 21 31; do FASTKV_FUSED_CONVEYOR=$d FASTKV_FUSED_TUNE=0 timeout 120 python tools/exp_occ3.py 2>&1 | grep -v amdgpu | tail -1; done
