#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do for t in 0 1; do FASTKV_FUSED_TUNE=$t python bench.py --no-extras 2>/dev/null | python -c "
import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TUNE=$t', j['ms_per_step'])"; done; done
export EXP_B=8
for i in 1 2 3; do for t in 0 1; do FASTKV_FUSED_TUNE=$t python tools/exp_occ3.py 2>&1 | grep -v amdgpu | tail -1 | cut -c1-90; done; done
