#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for ppc in 0 22 32 11; do
python bench.py --config 4 --steps 200 --no-cpu-baseline --no-side --no-dense --pages-per-chunk $ppc > $O/r4k_cfg4_ppc$ppc.json 2> $O/r4k_cfg4_ppc$ppc.err || exit 1
done
python scripts/summ.py $O/r4k_*.json
python scripts/wallstamps.py --config 4 > $O/r4k_wall_cfg4.log 2>&1; head -12 $O/r4k_wall_cfg4.log
