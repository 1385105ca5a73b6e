#!/bin/bash
# usage (on the GPU box): tools/prof_kernels.sh <tag> -- per-kernel rocprofv3 durations of dense update_kv calls at S=32768 and S=2048
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kp_$tag -- python $GRAFT_REPO_ROOT/tools/bench_update.py > $GRAFT_REPO_ROOT/gpurun_out/kp_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob, collections, statistics as st
f=glob.glob('gpurun_out/kp_$tag/*/*kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if 'fk::' in r['Kernel_Name']:
        agg[(r['Kernel_Name'].split('(')[0][-34:], r['Grid_Size_X']+'x'+r['Grid_Size_Y']+'x'+r['Grid_Size_Z'])].append(int(r['End_Timestamp'])-int(r['Start_Timestamp']))
for k,v in sorted(agg.items()): print(f"{k[0]:36s} grid={k[1]:16s} n={len(v):5d} median={st.median(v)/1e3:7.1f}us min={min(v)/1e3:7.1f}us")
PY
