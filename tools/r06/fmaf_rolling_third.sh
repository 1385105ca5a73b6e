#!/bin/bash
# round 6, third contact: the explicit hand-over of the rolling launch's record areas ("done" granules): the slow-entry test, the rolling
# suite, a soak under FASTKV_CONTRACTION=fmaf (placement policy "count"), the same soak under mfma16, the bench line
cd $GRAFT_REPO_ROOT
( time timeout 1500 python -m pytest tests/test_rolling_gpu.py -q -m gpu -x ) 2>&1 | tail -n 8 | tee gpurun_out/r06c_rolling_tests_fmaf.log
FASTKV_STRICT_PLACEMENT=0 timeout 500 python tools/soak_rolling.py 300 63 > gpurun_out/r06c_soak_fmaf_full.log 2>&1; tail -n 4 gpurun_out/r06c_soak_fmaf_full.log | tee gpurun_out/r06c_soak_fmaf.log
FASTKV_CONTRACTION=mfma16 timeout 400 python tools/soak_rolling.py 200 64 > gpurun_out/r06c_soak_mfma16_full.log 2>&1; tail -n 3 gpurun_out/r06c_soak_mfma16_full.log | tee gpurun_out/r06c_soak_mfma16.log
for r in 0 1; do FASTKV_FUSED_ROLLING=$r python tools/exp_interleave.py 2>&1 | grep "B=8\|B=16"; done | tee gpurun_out/r06c_interleave.log
python bench.py --no-ttft > gpurun_out/r06c_bench.json 2> gpurun_out/r06c_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06c_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['contraction'], json.dumps(d.get('roofline'))[:700])
print(json.dumps(d.get('kernels')), d.get('step_ms_by_contract'))
PY
