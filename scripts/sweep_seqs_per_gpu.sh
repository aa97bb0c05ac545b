#!/bin/bash
# Sequences per GPU 1 .. 16 at BASELINE configs[2] (32K context, budget 2048): where the batched launches cross the 7x / 0.6
# north-star bar.  One bench.py line per count -> gpurun_out/<round>_seqs_per_gpu_sweep.jsonl + a table on stdout.
#   scripts/sweep_seqs_per_gpu.sh r05
R=${1:-r05}
O=gpurun_out/${R}_seqs_per_gpu_sweep.jsonl; mkdir -p gpurun_out; : > $O
for n in 1 2 3 4 5 6 8 12 16; do
  timeout -k 10 300 python bench.py --seqs-per-gpu $n --steps 100 --warmup 10 --no-side --no-cpu-baseline 2> gpurun_out/${R}_sweep_$n.err | grep '^{' >> $O || { tail -3 gpurun_out/${R}_sweep_$n.err; exit 1; }
done
python3 - "$O" <<'PY'
import json, sys
print("| sequences per GPU | launches per layer | µs per sequence-layer | chain fraction of 8 TB/s | vs full-KV (same batch) | vs full-KV (one sequence) | tokens/s |")
print("|---|---|---|---|---|---|---|")
for l in open(sys.argv[1]):
    d = json.loads(l)
    print(f"| {d['config']['sequences_per_gpu']} | {d['config']['launches_per_layer'][:1]} | {d['selfattn_us_per_layer']:.2f} | {d['chain_frac_of_hbm_peak']:.3f} | "
          f"{(d.get('speedup_vs_batched_dense') or d.get('speedup_vs_dense') or 0):.2f} | {d.get('speedup_vs_dense') or 0:.2f} | {d['value']:.0f} |")
PY
