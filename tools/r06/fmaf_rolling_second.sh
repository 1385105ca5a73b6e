#!/bin/bash
# round 6, second contact: the fma-chain kernel with the mixed-fma trims, rolling on / off; the 16 post-TSP layers under other launch geometries
# (experiments build in build_x_exp); the rolling tests and a soak under FASTKV_CONTRACTION=fmaf with the placement policy on "count"
cd $GRAFT_REPO_ROOT
export FASTKV_CONTRACTION=fmaf
out=gpurun_out/r06b_fmaf_rolling.log
: > $out
for r in 0 1; do FASTKV_FUSED_ROLLING=$r python tools/exp_interleave.py 2>&1 | grep "B=8\|B=16" >> $out; done
echo "--- post-TSP block (16 x 2048 tokens), experiments build" >> $out
X=$GRAFT_REPO_ROOT/build_x_exp
for c in fmaf mfma16; do
  for v in "" "FASTKV_FUSED_MAX_WGS=256" "FASTKV_FUSED_ROLLING_PERT=2" "FASTKV_FUSED_ROLLING_PERT=1" "FASTKV_FUSED_ROLLING_PERT=2 FASTKV_FUSED_ROLLING_F=4" "FASTKV_FUSED_ROLLING_PERT=1 FASTKV_FUSED_ROLLING_F=4"; do
    echo -n "$c $v :: " >> $out
    env FASTKV_CONTRACTION=$c FASTKV_BUILD_DIR=$X $v python tools/exp_small_layers.py 2>&1 | grep "us per call" >> $out
  done
done
cat $out
( time timeout 1200 python -m pytest tests/test_rolling_gpu.py -q -m gpu -x ) 2>&1 | tail -n 6 | tee gpurun_out/r06b_rolling_tests_fmaf.log
FASTKV_STRICT_PLACEMENT=0 timeout 400 python tools/soak_rolling.py 240 62 2>&1 | tail -n 3 | tee gpurun_out/r06b_soak_fmaf.log
python bench.py --no-ttft > gpurun_out/r06b_bench_fmaf.json 2> gpurun_out/r06b_bench_fmaf.err
cut -c1-200 gpurun_out/r06b_bench_fmaf.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06b_bench_fmaf.json').read().strip().splitlines()[-1])
print(json.dumps(d.get('kernels')), d.get('step_ms_by_contract'))
PY
