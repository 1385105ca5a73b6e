#!/bin/bash
# Produces the round's measurement artefacts under gpurun_out/ (copy the summaries into profiles/):
#   bench JSON line, rocprofv3 --kernel-trace --stats of the same command, FETCH_SIZE / WRITE_SIZE PMC passes.
tag=${1:-r03}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python $R/bench.py > $R/gpurun_out/${tag}_bench.json 2> $R/gpurun_out/${tag}_bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_trace -- python3 $R/bench.py > $R/gpurun_out/${tag}_trace_bench.json 2> $R/gpurun_out/${tag}_trace.log
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2> $R/gpurun_out/${tag}_pmc_fetch.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/${tag}_pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-extras > /dev/null 2> $R/gpurun_out/${tag}_pmc_write.log
cd $R
python - <<PY
import csv, glob, collections, statistics as st, json
tag="$tag"
def med(path, counter):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name']==counter and 'fk::' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('(')[0].replace('void ',''), r['Grid_Size'])].append(float(r['Counter_Value']))
    return {k:(st.median(v),len(v)) for k,v in agg.items()}
out={}
for name,counter in (("fetch","FETCH_SIZE"),("write","WRITE_SIZE")):
    f=glob.glob(f'gpurun_out/{tag}_pmc_{name}/*/*counter_collection.csv')
    if f: out[counter]={f"{k[0]} grid={k[1]}":{"median":v[0],"n":v[1]} for k,v in med(f[0],counter).items()}
json.dump(out, open(f'gpurun_out/{tag}_pmc_summary.json','w'), indent=1)
print(json.dumps(out, indent=1)[:3000])
PY
bash tools/pmc_mfma.sh ${tag} > /dev/null 2>&1
python tools/profile_summarise.py ${tag} > /dev/null; cat gpurun_out/${tag}_bench.json | cut -c1-400
