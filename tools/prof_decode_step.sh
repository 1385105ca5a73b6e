#!/bin/bash
# usage (on the GPU box): tools/prof_decode_step.sh <tag> [bench_decode_step.py args] -- rocprofv3 durations of the decode step-attention launches
tag=${1:-x}; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ds_$tag -- python3 $GRAFT_REPO_ROOT/tools/bench_decode_step.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/ds_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
tail -2 gpurun_out/ds_$tag.log
python - <<PY
import csv, glob, collections, statistics as st
f=glob.glob('gpurun_out/ds_$tag/*/*kernel_trace.csv')[0]
agg=collections.defaultdict(list); rows=[r for r in csv.DictReader(open(f))]
for r in rows:
    if 'fk::' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0][-40:], r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items()): print(f"{k[0]:42s} grid={k[1]:16s} n={len(v):5d} median={st.median(v)/1e3:7.2f}us min={min(v)/1e3:7.2f}us")
# gaps between consecutive step launches (end -> next start)
ks=sorted([(int(r['Start_Timestamp']),int(r['End_Timestamp'])) for r in rows if 'decode_step' in r['Kernel_Name']])
gaps=[b[0]-a[1] for a,b in zip(ks,ks[1:]) if 0 <= b[0]-a[1] < 20000]
if gaps: print(f"gap between consecutive launches: median {st.median(gaps)/1e3:.2f} us (n={len(gaps)})")
PY
