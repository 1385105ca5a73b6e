#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_hip_parity.py tests/test_rolling_gpu.py -x -q -m gpu -k "many_heads or switches" ) 2>&1 | grep -v "^$" | tail -n 6
python tools/exp_recipe_128k.py 2>&1 | grep -v amdgpu | tail -n 3
