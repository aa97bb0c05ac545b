#!/bin/bash
# Sequences per GPU 1 .. 16 at BASELINE configs[2] (32K context, budget 2048): where the batched launches cross the 7x / 0.6
# north-star bar.  One bench.py line per count -> gpurun_out/r04_seqs_sweep.jsonl + a table on stdout.
O=gpurun_out/r04_seqs_sweep.jsonl; : > $O
for n in 1 2 3 4 6 8 12 16; do
  timeout -k 10 300 python bench.py --seqs-per-gpu $n --steps 100 --warmup 10 --no-side --no-cpu-baseline 2>/dev/null | grep '^{' >> $O || exit 1
done
python3 - <<'PY'
import json
print("| sequences per GPU | µs per sequence-layer | chain fraction of 8 TB/s | vs full-KV (same batch) | vs full-KV (one sequence) | tokens/s |")
print("|---|---|---|---|---|---|")
for l in open("gpurun_out/r04_seqs_sweep.jsonl"):
    d = json.loads(l)
    print(f"| {d['config']['sequences_per_gpu']} | {d['selfattn_us_per_layer']:.2f} | {d['chain_frac_of_hbm_peak']:.3f} | "
          f"{(d.get('speedup_vs_batched_dense') or d.get('speedup_vs_dense') or 0):.2f} | {d.get('speedup_vs_dense') or 0:.2f} | {d['value']:.0f} |")
PY
