#!/bin/bash
# round 6, seventh contact: pacing of the rolling launch (an entry starts phase A when the entry before it has published its row maxima).
# A/B through the experiments build (FASTKV_FUSED_TUNE: 1 = as before, 3 = paced), both contracts; the rolling suite; per-entry stamps;
# a short soak; the bench line
cd $GRAFT_REPO_ROOT
X=$GRAFT_REPO_ROOT/build_x_exp
out=gpurun_out/r06g_pacing_ab.log
: > $out
for c in fmaf mfma16; do for i in 1 2; do for t in 1 3; do
  echo -n "$c TUNE=$t :: " >> $out
  FASTKV_CONTRACTION=$c FASTKV_BUILD_DIR=$X FASTKV_FUSED_TUNE=$t python tools/exp_interleave.py 2>&1 | grep "B=8\|B=16" | tr '\n' ' ' >> $out; echo >> $out
done; done; done
cat $out
( timeout 1500 python -m pytest tests/test_rolling_gpu.py -q -m gpu -x 2>&1 | tail -n 4 ) | tee gpurun_out/r06g_rolling_tests.log
( FASTKV_CONTRACTION=mfma16 timeout 1500 python -m pytest tests/test_rolling_gpu.py -q -m gpu -x 2>&1 | tail -n 4 ) | tee gpurun_out/r06g_rolling_tests_mfma16.log
export FASTKV_CXXFLAGS=-DFK_STAMP
( FASTKV_BUILD_DIR=$GRAFT_REPO_ROOT/build_x_stamp python tools/stamp_rolling.py 8 2>&1 | grep -v amdgpu.ids ) > gpurun_out/r06g_stamps_rolling_fmaf_paced.log
unset FASTKV_CXXFLAGS
head -n 24 gpurun_out/r06g_stamps_rolling_fmaf_paced.log
( FASTKV_STRICT_PLACEMENT=0 timeout 400 python tools/soak_rolling.py 240 66 2>&1 | grep -v "RuntimeWarning\|raise_if_aborted()\|amdgpu.ids" | tail -n 6 ) | tee gpurun_out/r06g_soak_fmaf_paced.log
python bench.py --no-ttft > gpurun_out/r06g_bench.json 2> gpurun_out/r06g_bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06g_bench.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['contraction'], d['roofline']['avg_launch_us'], d['roofline']['frac'], json.dumps(d.get('kernels')), d.get('step_ms_by_contract'), d['published_recipe']['ms_per_step'])
PY
