#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python tools/exp_abort_threshold.py > gpurun_out/r05c_abort_threshold.log 2>&1
( time python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "many_heads or half_the_chip or wide_sweep" ) > gpurun_out/r05c_tests.log 2>&1
python bench.py --no-ttft --no-legs > gpurun_out/r05c_bench.json 2> gpurun_out/r05c_bench.err
cat gpurun_out/r05c_abort_threshold.log | grep -v amdgpu; tail -n 4 gpurun_out/r05c_tests.log; python - <<'PY'
import json
j=json.loads(open('gpurun_out/r05c_bench.json').read().strip().splitlines()[-1])
print(j['ms_per_step'], {k:v['us_per_step'] for k,v in j['kernels'].items()})
r=j['published_recipe']; print('recipe', r['ms_per_step'], {k:(v['launches_per_step'],v['avg_us']) for k,v in r['kernels'].items()}, r['layer_by_layer'])
PY
