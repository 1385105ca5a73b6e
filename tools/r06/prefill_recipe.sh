#!/bin/bash
# round 6: python -m benchmark.prefill with the reference's published flags under the new default contract (random-init geometries)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r06_prefill_recipe.log
echo "# python -m benchmark.prefill with the reference's published flags (scripts/eval_prefill.sh:4-12; scripts2/eval_prefill.sh:37-47), random-init geometries, MI355X, default contract = fp32 fma chain (round 6), 1 warm-up + 5 runs (fullkv: 3)" > $out
for m in "llama3-8b --tsp_idx 15" "ministral-8b --tsp_idx 17"; do
  set -- $m
  ( timeout 900 python -m benchmark.prefill --method fastkv --model_name $1 $2 $3 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode proportional --num_warmups 1 --num_runs 5 2>&1 | grep "^\[prefill\]" | sed "s/^/$1 $2 $3 --tsp_rate 0.2 --retain_rate 0.1 --eviction_mode proportional: /" ) >> $out
done
( timeout 900 python -m benchmark.prefill --method fastkv --model_name llama3-8b --max_capacity_prompts 2048 --num_warmups 1 --num_runs 5 2>&1 | grep "^\[prefill\]" | sed "s/^/llama3-8b --max_capacity_prompts 2048 (constant budget): /" ) >> $out
( timeout 900 python -m benchmark.prefill --method fullkv --model_name llama3-8b --num_warmups 1 --num_runs 3 2>&1 | grep "^\[prefill\]" | sed "s/^/llama3-8b --method fullkv: /" ) >> $out
cut -c1-200 $out
