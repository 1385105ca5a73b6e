#!/bin/bash
# round 6, tenth contact: the random stress tools on the shipping library under the new default contract and under mfma16
cd $GRAFT_REPO_ROOT
( timeout 900 python tools/stress_parity.py 1500 76 2>&1 | grep -v amdgpu.ids | tail -n 4 ) | tee gpurun_out/r06j_stress_fmaf.log
( FASTKV_CONTRACTION=mfma16 timeout 900 python tools/stress_parity.py 700 77 2>&1 | grep -v amdgpu.ids | tail -n 4 ) | tee gpurun_out/r06j_stress_mfma16.log
( timeout 900 python tools/stress_dist.py 48 78 2>&1 | grep -v amdgpu.ids | tail -n 3 ) | tee gpurun_out/r06j_stress_dist.log
( SOAK_SMAX=131072 SOAK_BMAX=6 SOAK_SMIN=40000 FASTKV_STRICT_PLACEMENT=0 timeout 400 python tools/soak_rolling.py 200 79 2>&1 | grep -v "RuntimeWarning\|raise_if_aborted()\|amdgpu.ids" | tail -n 3 ) | tee gpurun_out/r06j_soak_long_rows_fmaf.log
