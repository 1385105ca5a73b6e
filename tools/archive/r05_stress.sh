#!/bin/bash
# round 5, end: the random stress tools on the final library (new code paths: rank_group<2/1>, the TSP fold, FUSED_MAX_WGS 2048)
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_rolling_gpu.py -x -q -m gpu -k "switches" ) 2>&1 | tail -n 4
( time python tools/stress_parity.py 1200 51 ) > gpurun_out/r05_stress_parity.log 2>&1; tail -n 3 gpurun_out/r05_stress_parity.log
( time STRESS_ENTRIES_P=1 python tools/stress_parity.py 600 52 ) > gpurun_out/r05_stress_parity_entries.log 2>&1; tail -n 3 gpurun_out/r05_stress_parity_entries.log
( time python tools/soak_rolling.py 150 7 ) > gpurun_out/r05_soak_rolling.log 2>&1; tail -n 2 gpurun_out/r05_soak_rolling.log
( time python tools/stress_dist.py ) > gpurun_out/r05_stress_dist.log 2>&1; tail -n 2 gpurun_out/r05_stress_dist.log
