#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for steps in 300 1000; do
python bench.py --config 2 --steps $steps --no-cpu-baseline --no-side > $O/r4i_cfg2_fused_$steps.json 2> $O/r4i_cfg2_fused_$steps.err || exit 1
python bench.py --config 2 --steps $steps --no-cpu-baseline --no-side --separate-dense-append > $O/r4i_cfg2_sep_$steps.json 2> $O/r4i_cfg2_sep_$steps.err || exit 1
(cd build/r03tree && python bench.py --config 2 --steps $steps --no-cpu-baseline --no-side > $O/r4i_cfg2_r03_$steps.json 2> $O/r4i_cfg2_r03_$steps.err) || exit 1
done
python scripts/summ.py $O/r4i_*.json
