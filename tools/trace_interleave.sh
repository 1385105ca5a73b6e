#!/bin/bash
# usage (GPU box): tools/trace_interleave.sh -- kernel start/end times of the fused launches of the interleaved schedule (do they overlap?)
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/il -- python3 $GRAFT_REPO_ROOT/tools/exp_interleave.py > $GRAFT_REPO_ROOT/gpurun_out/il.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<PY
import csv, glob
f=glob.glob('gpurun_out/il/*/*kernel_trace.csv')[0]
rows=[r for r in csv.DictReader(open(f)) if 'score_fused' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-16:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    print(f"queue {r.get('Queue_Id','?'):>3} grid {r['Grid_Size_X']}x{r['Grid_Size_Y']}  start {(int(r['Start_Timestamp'])-t0)/1e3:8.1f}  end {(int(r['End_Timestamp'])-t0)/1e3:8.1f}  dur {(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:6.1f} us")
PY
