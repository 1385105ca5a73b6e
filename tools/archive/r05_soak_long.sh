#!/bin/bash
# round 5: the long soaks on the end-of-round library (what found three latent-bug classes in rounds 3 and 4)
cd $GRAFT_REPO_ROOT
b=${1:-20}   # seed base (the first run of the round: 20)
( time python tools/soak_rolling.py 600 $((b+1)) ) > gpurun_out/r05_soak_rolling_600a.log 2>&1; tail -n 2 gpurun_out/r05_soak_rolling_600a.log | head -1
( time SOAK_SMAX=${SOAK2_SMAX:-32768} SOAK_BMAX=${SOAK2_BMAX:-20} SOAK_SMIN=${SOAK2_SMIN:-8192} python tools/soak_rolling.py 400 $((b+2)) ) > gpurun_out/r05_soak_rolling_400b.log 2>&1; tail -n 2 gpurun_out/r05_soak_rolling_400b.log | head -1
( time python tools/stress_parity.py 5000 $((b+41)) ) > gpurun_out/r05_stress_parity_5000.log 2>&1; grep "cases," gpurun_out/r05_stress_parity_5000.log
( time STRESS_ENTRIES_P=1 python tools/stress_parity.py 1500 $((b+42)) ) > gpurun_out/r05_stress_parity_entries_1500.log 2>&1; grep "cases," gpurun_out/r05_stress_parity_entries_1500.log
( time python tools/stress_decode.py 600 $((b-15)) ) > gpurun_out/r05_stress_decode.log 2>&1; tail -n 4 gpurun_out/r05_stress_decode.log | head -1
