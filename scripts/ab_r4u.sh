#!/bin/bash
# Round 4, VERDICT item 4 (i): the batched attention launch with 2 workgroups per CU on half-heads (64 pages each; merge
# launch added) against the default one workgroup per head, 8 x cfg 3 and cfg 5.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for ppc in 0 64 32; do
python bench.py --config 3 --seqs-per-gpu 8 --steps 100 --no-cpu-baseline --no-side --no-dense --pages-per-chunk $ppc > $O/r4u_cfg3x8_ppc$ppc.json 2> $O/r4u_cfg3x8_ppc$ppc.err || exit 1
python bench.py --config 5 --steps 100 --no-cpu-baseline --no-side --no-dense --pages-per-chunk $ppc > $O/r4u_cfg5_ppc$ppc.json 2> $O/r4u_cfg5_ppc$ppc.err || exit 1
done
python scripts/summ.py $O/r4u_*.json
