#!/bin/bash
# does the finish stream's work run BESIDE the next scoring launch at all?  kernel trace of a few steps, overlap computed from the timestamps
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in 0 1; do
  export FASTKV_FINISH_STREAM=$m
  rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r05_fin_trace_$m -- python3 $R/tools/exp_finish_stream.py > $R/gpurun_out/r05_fin_trace_$m.log 2>&1
done
cd $R
python3 - <<'PY' 2>&1 | tee gpurun_out/r05_finish_trace.log
import csv, glob, statistics as st
for m in (0, 1):
    f = glob.glob(f'gpurun_out/r05_fin_trace_{m}/*/*kernel_trace.csv')
    rows = [r for r in csv.DictReader(open(f[0])) if 'fk::' in r['Kernel_Name']]
    ev = sorted(((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void fk::', ''), r.get('Grid_Size_X', r.get('Grid_Size', ''))) for r in rows))
    ev = ev[len(ev) // 2:]                                     # the timed steps
    fused = [e for e in ev if 'score_fused' in e[2]]
    other = [e for e in ev if 'score_fused' not in e[2]]
    def dur(name, grid=None):
        d = [(e[1] - e[0]) / 1e3 for e in ev if name in e[2] and (grid is None or e[3] == grid)]
        return (round(st.median(d), 1), len(d)) if d else None
    ov = 0
    for s0, s1, _, _ in fused:
        for o0, o1, _, _ in other:
            ov += max(0, min(s1, o1) - max(s0, o0))
    print(f"FINISH_STREAM={m}: kernels {len(ev)}; score_fused 65536x8 median {dur('score_fused', '65536')}, compact_kv {dur('compact_kv')}, select_split {dur('select_split')}, "
          f"rank_group {dur('rank_group')}; time other fk kernels spend inside a score_fused launch: {ov / 1e3 / max(1, len(fused)):.1f} us per fused launch")
PY
rm -rf gpurun_out/r05_fin_trace_0 gpurun_out/r05_fin_trace_1
