#!/bin/bash
# round 6, fifth contact: is the explicit hand-over of the record areas what round 6's first fma-chain soak (seed 62) ran into?  The same seed
# on an experiments build with the hand-over OFF, then with the product library; the tests that failed in the fourth run; who arrives last at
# the hand-offs (stamp build); the round's first profile
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/build_x_exp
( FASTKV_BUILD_DIR=$X FASTKV_FUSED_NO_HANDOVER=1 FASTKV_SPIN_LIMIT_MS=400 FASTKV_STRICT_PLACEMENT=0 timeout 300 python tools/soak_rolling.py 120 62 2>&1 | grep -v "^$" | grep -v "RuntimeWarning\|raise_if_aborted()\|amdgpu.ids" | head -n 60 ) > gpurun_out/r06e_soak_no_handover.log 2>&1
echo "== hand-over OFF (experiments build), seed 62:"; grep -c REPORTED gpurun_out/r06e_soak_no_handover.log; tail -n 2 gpurun_out/r06e_soak_no_handover.log | cut -c1-300
( FASTKV_STRICT_PLACEMENT=0 timeout 400 python tools/soak_rolling.py 240 62 2>&1 | grep -v "RuntimeWarning\|raise_if_aborted()\|amdgpu.ids" | tail -n 20 ) > gpurun_out/r06e_soak_handover.log 2>&1
echo "== hand-over ON (product library), seed 62:"; tail -n 2 gpurun_out/r06e_soak_handover.log | cut -c1-300
( timeout 900 python -m pytest tests/test_hip_parity.py tests/test_stress_gpu.py -q -m gpu -k "arithmetic_contract or random_parity or half_the_chip or abandoned" 2>&1 | tail -n 8 ) | tee gpurun_out/r06e_failed_tests.log
S=$GRAFT_REPO_ROOT/build_x_stamp
( FASTKV_BUILD_DIR=$S python tools/stamp_arrivals.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06e_arrivals_fmaf.log
( FASTKV_BUILD_DIR=$S FASTKV_CONTRACTION=mfma16 python tools/stamp_arrivals.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06e_arrivals_mfma16.log
tail -n 4 gpurun_out/r06e_arrivals_fmaf.log; tail -n 4 gpurun_out/r06e_arrivals_mfma16.log
bash tools/profile_round.sh r06a > gpurun_out/r06a_profile_round.log 2>&1; tail -n 3 gpurun_out/r06a_profile_round.log | cut -c1-600
