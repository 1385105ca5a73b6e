#!/bin/bash
cd $GRAFT_REPO_ROOT
( time python -m pytest tests/test_hip_parity.py tests/test_model_parity_gpu.py tests/test_wiring_gpu.py tests/test_stress_gpu.py -x -q -m gpu ) > gpurun_out/r05g_tests.log 2>&1
tail -n 5 gpurun_out/r05g_tests.log
show() { python - "$1" <<'PY'
import json, sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], j['ms_per_step'], {k:(v['launches_per_step'], v['avg_us']) for k,v in j['kernels'].items()}, 'lbl', j['layer_by_layer']['ms_per_step'], 'recipe', j['published_recipe']['ms_per_step'], j['published_recipe']['layer_by_layer']['ms_per_step'])
PY
}
FASTKV_TSP_FOLD=0 python bench.py --no-ttft --no-legs --no-cpu-baseline > gpurun_out/r05g_bench_nofold.json 2>/dev/null; show gpurun_out/r05g_bench_nofold.json
python bench.py --no-ttft --no-legs --no-cpu-baseline > gpurun_out/r05g_bench_fold.json 2>/dev/null; show gpurun_out/r05g_bench_fold.json
FASTKV_FUSED_MAX_WGS=256 python bench.py --no-ttft --no-legs --no-cpu-baseline > gpurun_out/r05g_bench_wgs256.json 2>/dev/null; show gpurun_out/r05g_bench_wgs256.json
