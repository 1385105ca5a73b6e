#!/bin/bash
cd $GRAFT_REPO_ROOT
show() { python - "$1" <<'PY'
import json, sys
j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=j['published_recipe']
print(sys.argv[1], j['ms_per_step'], 'recipe', r['ms_per_step'], {k:(v['launches_per_step'], v['avg_us']) for k,v in r['kernels'].items()})
PY
}
FASTKV_DEFER_MAX_LEN=8192 python bench.py --no-ttft --no-legs --no-cpu-baseline > gpurun_out/r05h_bench_maxlen8k.json 2>/dev/null; show gpurun_out/r05h_bench_maxlen8k.json
# the reference's published recipe through the TTFT harness, the script's own flags (scripts/eval_prefill.sh:4-12; scripts2/eval_prefill.sh:37-47)
( time python -m benchmark.prefill --method fastkv --model_path llama3-8b --tsp_idx 15 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode proportional --num_warmups 1 --num_runs 5 --save_txt "" ) > gpurun_out/r05h_prefill_recipe_llama.log 2>&1
( time python -m benchmark.prefill --method fullkv --model_path llama3-8b --num_warmups 1 --num_runs 3 --save_txt "" ) > gpurun_out/r05h_prefill_fullkv_llama.log 2>&1
( time python -m benchmark.prefill --method fastkv --model_path ministral-8b --tsp_idx 17 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode proportional --num_warmups 1 --num_runs 5 --save_txt "" ) > gpurun_out/r05h_prefill_recipe_ministral.log 2>&1
grep "\[prefill\]\|real" gpurun_out/r05h_prefill_*.log
