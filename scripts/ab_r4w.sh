#!/bin/bash
# Round 4 experiment: 8 sequences per GPU as G groups of 8/G batched sequences, each group on its own stream inside one hipGraph.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for g in 1 2 4; do
python bench.py --config 3 --seqs-per-gpu 8 --seq-groups $g --steps 100 --no-cpu-baseline --no-side --no-dense > $O/r4w_cfg3x8_g$g.json 2> $O/r4w_cfg3x8_g$g.err || { tail -3 $O/r4w_cfg3x8_g$g.err; exit 1; }
python bench.py --config 5 --seq-groups $g --steps 100 --no-cpu-baseline --no-side --no-dense > $O/r4w_cfg5_g$g.json 2> $O/r4w_cfg5_g$g.err || { tail -3 $O/r4w_cfg5_g$g.err; exit 1; }
done
python scripts/summ.py $O/r4w_*.json
