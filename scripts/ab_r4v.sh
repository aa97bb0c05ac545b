#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for ppc in 0 16 11 13; do
python bench.py --config 3 --steps 300 --no-cpu-baseline --no-side --no-dense --pages-per-chunk $ppc > $O/r4v_cfg3_ppc$ppc.json 2> $O/r4v_cfg3_ppc$ppc.err || exit 1
done
python scripts/summ.py $O/r4v_*.json
python scripts/wallstamps.py --ppc 16 > $O/r4v_wall_ppc16.log 2>&1; head -9 $O/r4v_wall_ppc16.log
