#!/bin/bash
cd $GRAFT_REPO_ROOT
export EXP_B=16 FASTKV_FUSED_CONVEYOR=1
for t in 0 2 4 6 8 16 32 56 62; do FASTKV_FUSED_TUNE=$t FASTKV_SPIN_LIMIT_MS=200 timeout 120 python tools/exp_occ3.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-200; done
