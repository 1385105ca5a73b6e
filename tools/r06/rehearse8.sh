#!/bin/bash
# round 6: the N = 8 launcher rehearsal on a one-GPU box (8 ranks share the GPU, gloo rendezvous, no-wait kernels) and the ONE RCCL rank a
# one-GPU box allows (BENCH_BACKEND=nccl python bench.py --gpus 1) -- VERDICT r05 next #6.  Writes profiles-ready JSON into gpurun_out/.
cd $GRAFT_REPO_ROOT
t0=$(date +%s.%N)
BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 8 --steps 10 --warmup 2 > gpurun_out/r06_bench8_gloo.out 2> gpurun_out/r06_bench8_gloo.err
t1=$(date +%s.%N)
python - <<PY
import json
out=[l for l in open('gpurun_out/r06_bench8_gloo.out').read().splitlines() if l.startswith('{')]
err=open('gpurun_out/r06_bench8_gloo.err').read()
legs=[json.loads(l[len('LEGS_JSON '):]) for l in err.splitlines() if l.startswith('LEGS_JSON ')]
line=json.loads(out[-1])
json.dump({"command": "BENCH_BACKEND=gloo python bench.py --gpus 8 --steps 10 --warmup 2 (8 ranks sharing one MI355X, launched by bench.py itself)",
           "wall_time_s_of_the_whole_command": round($t1-$t0,1), "contract_line_printed_after_s_of_rank_start": line.get("line_after_s"),
           "stdout_lines": len(out), "line": line, "legs_after_the_line_from_stderr": legs}, open('gpurun_out/r06_bench8_gloo_rehearsal.json','w'), indent=1)
print("N=8 line after", line.get("line_after_s"), "s; ranks_seen", line.get("ranks_seen"), "roofline" in line, "whole command", round($t1-$t0,1), "s")
PY
BENCH_BACKEND=nccl timeout 900 python bench.py --gpus 1 --steps 10 --warmup 2 --no-ttft > gpurun_out/r06_bench1_nccl.json 2> gpurun_out/r06_bench1_nccl.err
python - <<PY
import json
l=[x for x in open('gpurun_out/r06_bench1_nccl.json').read().splitlines() if x.startswith('{')]
d=json.loads(l[-1]); print("one RCCL rank:", d.get("backend"), d.get("ms_per_step"), [k for k in d if k in ("seq_sharded_weak","seq_sharded_128k","tp","sp_ttft_128k")])
PY
