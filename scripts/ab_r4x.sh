#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for g in 2 4; do
python bench.py --config 3 --seqs-per-gpu 8 --seq-groups $g --pages-per-chunk 128 --steps 100 --no-cpu-baseline --no-side --no-dense > $O/r4x_cfg3x8_g${g}_ppc128.json 2> $O/r4x_cfg3x8_g${g}.err || { tail -3 $O/r4x_cfg3x8_g$g.err; exit 1; }
python bench.py --config 5 --seq-groups $g --pages-per-chunk 128 --steps 100 --no-cpu-baseline --no-side --no-dense > $O/r4x_cfg5_g${g}_ppc128.json 2> $O/r4x_cfg5_g$g.err || { tail -3 $O/r4x_cfg5_g$g.err; exit 1; }
done
python scripts/summ.py $O/r4x_*.json
