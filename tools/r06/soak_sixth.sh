#!/bin/bash
# round 6, sixth contact: the long soak of the rolling launch under the fma chain (>= 1 M groups with the round's other
# soaks), the known trigger in the rolling geometry x 300, round 3's reproducers and the self-test under the new default; per-entry stage
# stamps and who arrives last at the hand-offs (stamp build); the 8-rank rehearsal and the one RCCL rank
cd $GRAFT_REPO_ROOT
export FASTKV_SELFTEST=1
( FASTKV_STRICT_PLACEMENT=0 timeout 1300 python tools/soak_rolling.py 1100 65 2>&1 | grep -v "RuntimeWarning\|raise_if_aborted()\|amdgpu.ids" | tail -n 12 ) > gpurun_out/r06f_soak_fmaf_long.log 2>&1; tail -n 2 gpurun_out/r06f_soak_fmaf_long.log | cut -c1-250
( timeout 900 python tools/repro_rolling_slow.py 100 2>&1 | grep -v amdgpu.ids | tail -n 8 ) | tee gpurun_out/r06f_repro_rolling_slow.log
( SPECIAL=Q timeout 600 python tools/repro_race.py 100 2>&1 | grep -v amdgpu.ids | tail -n 3 ) | tee gpurun_out/r06f_repro_race.log
( timeout 900 python -m pytest tests/test_hip_parity.py -q -m gpu -k "known_trigger or slow_entry_does_not_disturb or selftest" 2>&1 | tail -n 3 ) | tee gpurun_out/r06f_known_trigger.log
( for i in 1 2 3 4 5; do timeout 300 python -c "
from fastkv_amd import selftest
print('co_residency self-test: launches that differ =', selftest.co_residency(launches=100))" 2>&1 | grep -v amdgpu.ids | tail -n 1; done ) | tee gpurun_out/r06f_selftest.log
unset FASTKV_SELFTEST
S=$GRAFT_REPO_ROOT/build_x_stamp
export FASTKV_CXXFLAGS=-DFK_STAMP
( FASTKV_BUILD_DIR=$S python tools/stamp_rolling.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f_stamps_rolling_fmaf.log
( FASTKV_BUILD_DIR=$S python tools/stamp_arrivals.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f_arrivals_fmaf.log
( FASTKV_BUILD_DIR=$S FASTKV_CONTRACTION=mfma16 python tools/stamp_arrivals.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06f_arrivals_mfma16.log
unset FASTKV_CXXFLAGS
tail -n 12 gpurun_out/r06f_stamps_rolling_fmaf.log; tail -n 4 gpurun_out/r06f_arrivals_fmaf.log; tail -n 4 gpurun_out/r06f_arrivals_mfma16.log
bash tools/r06/rehearse8.sh 2>&1 | tail -n 4
