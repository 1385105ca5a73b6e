#!/bin/bash
# usage (on the GPU box): tools/pmc_mfma.sh <tag> -- matrix-pipe / vector-ALU activity counters of the library's kernels over a short bench run
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/${tag}_pmc_mfma -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2> $R/gpurun_out/${tag}_pmc_mfma.log
cd $R
python - <<PY
import csv, glob, collections, statistics as st, json
f=glob.glob('gpurun_out/${tag}_pmc_mfma/*/*counter_collection.csv')
out={}
if f:
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        if 'fk::' in r['Kernel_Name']:
            agg[r['Kernel_Name'].split('(')[0].replace('void ','')+' grid='+r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k,c in agg.items(): out[k]={n:st.median(v) for n,v in c.items()}
json.dump(out, open('gpurun_out/${tag}_pmc_mfma_summary.json','w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
