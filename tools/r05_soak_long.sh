#!/bin/bash
# round 5: the long soaks on the end-of-round library (what found three latent-bug classes in rounds 3 and 4)
cd $GRAFT_REPO_ROOT
( time python tools/soak_rolling.py 600 21 ) > gpurun_out/r05_soak_rolling_600a.log 2>&1; tail -n 2 gpurun_out/r05_soak_rolling_600a.log | head -1
( time python tools/soak_rolling.py 400 22 ) > gpurun_out/r05_soak_rolling_400b.log 2>&1; tail -n 2 gpurun_out/r05_soak_rolling_400b.log | head -1
( time python tools/stress_parity.py 5000 61 ) > gpurun_out/r05_stress_parity_5000.log 2>&1; grep "cases," gpurun_out/r05_stress_parity_5000.log
( time STRESS_ENTRIES_P=1 python tools/stress_parity.py 1500 62 ) > gpurun_out/r05_stress_parity_entries_1500.log 2>&1; grep "cases," gpurun_out/r05_stress_parity_entries_1500.log
( time python tools/stress_decode.py 600 5 ) > gpurun_out/r05_stress_decode.log 2>&1; tail -n 4 gpurun_out/r05_stress_decode.log | head -1
