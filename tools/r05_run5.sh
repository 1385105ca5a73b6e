#!/bin/bash
cd $GRAFT_REPO_ROOT
export EXP_B=8
for t in 0 1; do FASTKV_FUSED_TUNE=$t python tools/exp_occ3.py; done 2>&1 | grep -v amdgpu
export EXP_B=16
for t in 0 1; do FASTKV_FUSED_TUNE=$t python tools/exp_occ3.py; done 2>&1 | grep -v amdgpu
FASTKV_FUSED=0 python bench.py --no-ttft --no-legs --no-cpu-baseline > gpurun_out/r05e_bench_nowait.json 2> gpurun_out/r05e_bench_nowait.err
python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05e_bench_nowait.json').read().strip().splitlines()[-1])
print('no-wait step', j['ms_per_step'], {k:(v['launches_per_step'], v['avg_us'], v['us_per_step']) for k,v in j['kernels'].items()})
PY
python -m pytest tests/test_rolling_gpu.py -x -q -m gpu -k "abandoned" 2>&1 | tail -3
