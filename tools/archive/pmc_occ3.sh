#!/bin/bash
# counters behind the A/B of the rolling launch's two sizes (profiles/r05_occ3_*): SQ_WAIT_ANY / SQ_WAVE_CYCLES, vector issue
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export EXP_B=8
for o in 0 1; do
  export FASTKV_FUSED_OCC3=$o
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVES SQ_INSTS_MFMA --output-format csv -d $R/gpurun_out/r05d_pmc_occ3_$o -- python3 $R/tools/exp_occ3.py > $R/gpurun_out/r05d_pmc_occ3_$o.log 2>&1
  rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/r05d_pmc2_occ3_$o -- python3 $R/tools/exp_occ3.py > $R/gpurun_out/r05d_pmc2_occ3_$o.log 2>&1
done
cd $R
for o in 0 1; do echo "== OCC3=$o"; python tools/pmc_summary.py gpurun_out/r05d_pmc_occ3_$o | grep "kernel<"; python - <<PY
import csv, glob, collections, statistics as st
f=glob.glob('gpurun_out/r05d_pmc2_occ3_$o/*/*counter_collection.csv')
if f:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'score_fused' in r['Kernel_Name']: agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for g,c in agg.items(): print('grid',g,{n:st.median(v) for n,v in c.items()})
PY
done 2>&1 | tee gpurun_out/r05d_occ3_counters.log
rm -rf gpurun_out/r05d_pmc_occ3_0 gpurun_out/r05d_pmc_occ3_1 gpurun_out/r05d_pmc2_occ3_0 gpurun_out/r05d_pmc2_occ3_1
