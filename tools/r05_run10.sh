#!/bin/bash
cd $GRAFT_REPO_ROOT
for c in 0 1; do FASTKV_FUSED_CONVEYOR=$c timeout 300 python tools/exp_occ3.py; done 2>&1 | grep -v amdgpu
