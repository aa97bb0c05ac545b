#!/bin/bash
# Round 4, first GPU call: GPU test suite, then column-range ownership on / off (QUEST_TUNING=1 QUEST_COLRANGE=0) at cfg 3 and cfg 4.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r4b_tests.log 2>&1; echo "tests exit $?" | tee -a $O/r4b_tests.log
tail -5 $O/r4b_tests.log
for rep in 1 2; do
for cfg in 3 4; do
python bench.py --config $cfg --steps 300 --no-cpu-baseline --no-side > $O/r4b_cfg${cfg}_colrange_$rep.json 2> $O/r4b_cfg${cfg}_colrange_$rep.err || exit 1
QUEST_TUNING=1 QUEST_COLRANGE=0 python bench.py --config $cfg --steps 300 --no-cpu-baseline --no-side > $O/r4b_cfg${cfg}_slots_$rep.json 2> $O/r4b_cfg${cfg}_slots_$rep.err || exit 1
done
done
python scripts/summ.py $O/r4b_cfg*.json
