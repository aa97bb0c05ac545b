#!/bin/bash
# The reference's efficiency grid (scripts/bench_efficiency_e2e.sh:1-11 in the reference: token budgets 512 / 1024 / 2048 /
# 4096 / full x contexts 8192 / 16384 / 32768; paper Fig. 9 / 10, README.md:17) on the self-attention chain, Llama-2-7B MHA
# shapes, at 1 and 8 sequences per GPU.  One bench.py line per cell -> gpurun_out/<round>_budget_context_sweep.jsonl
# (copy it to profiles/; scripts/design_tables.py renders the table of BASELINE.md from it).
#   scripts/sweep_budget_context.sh r06 [layout]
R=${1:-r06}; LAY=${2:-NHD}
O=gpurun_out/${R}_budget_context_sweep.jsonl; mkdir -p gpurun_out; : > $O
for n in 1 8; do for ctx in 8192 16384 32768; do for budget in 512 1024 2048 4096 full; do
  b=$budget; [ $budget = full ] && b=1048576   # a budget that covers the cache: the reference's full-KV branch
  timeout -k 10 400 python bench.py --config 3 --seqlen $ctx --token-budget $b --seqs-per-gpu $n --steps 100 --warmup 10 \
      --layout $LAY --no-side --no-cpu-baseline 2> gpurun_out/${R}_bc_${n}_${ctx}_${budget}.err | grep '^{' >> $O \
      || { echo "FAILED: $n x $ctx, budget $budget"; tail -3 gpurun_out/${R}_bc_${n}_${ctx}_${budget}.err; }
  echo "done: $n sequence(s), context $ctx, budget $budget" >&2
done; done; done
python3 scripts/design_tables.py --budget-context $O
