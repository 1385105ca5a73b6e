#!/bin/bash
# usage (on the GPU box): tools/prof_e2e.sh <tag> -- rocprofv3 kernel trace of one benchmark.e2e run (graph-replayed decode over the slab cache);
# prints the kernels of the last 0.35 s of the trace (decode replays) by summed time
tag=${1:-x}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp FASTKV_SLAB_CACHE=1 PYTHONPATH=$R
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/e2e_$tag -- python3 -m benchmark.e2e --model_path llama3-8b --method fastkv --max_capacity_prompts 2048 --context_lengths 32768 --genlen 64 --num_warmups 0 --num_runs 1 --save_txt "" --random_tokens > $R/gpurun_out/e2e_$tag.log 2>&1
cd $R
grep "\[e2e\]" gpurun_out/e2e_$tag.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/e2e_$tag/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t_end = int(rows[-1]["End_Timestamp"])
cut = t_end - int(0.35e9)
agg = collections.defaultdict(lambda: [0, 0])
for r in rows:
    if int(r["Start_Timestamp"]) < cut:
        continue
    agg[r["Kernel_Name"][:90]][0] += 1
    agg[r["Kernel_Name"][:90]][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
tot = sum(v[1] for v in agg.values())
print(f"kernel time in the window {tot/1e6:.1f} ms of {(t_end-cut)/1e6:.1f} ms wall")
for n, (c, d) in sorted(agg.items(), key=lambda x: -x[1][1])[:22]:
    print(f"{d/1e6:8.2f} ms {c:6d} x {d/c/1e3:8.2f} us  {n}")
PY
