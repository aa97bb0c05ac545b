#!/bin/bash
# Round 4: full GPU suite on the current tree, then cfg 2 (dense layer in two launches vs the round-3 tree's three) and the
# batched side configurations with the padded page tables (variant 1) against the round-3 tree.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/r4f_tests.log 2>&1; echo "tests exit $?" | tee -a $O/r4f_tests.log
tail -4 $O/r4f_tests.log
for rep in 1 2; do
python bench.py --config 2 --steps 300 --no-cpu-baseline --no-side > $O/r4f_cfg2_new_$rep.json 2> $O/r4f_cfg2_new_$rep.err || exit 1
(cd build/r03tree && python bench.py --config 2 --steps 300 --no-cpu-baseline --no-side > $O/r4f_cfg2_r03_$rep.json 2> $O/r4f_cfg2_r03_$rep.err) || exit 1
done
python bench.py --config 3 --seqs-per-gpu 8 --steps 100 --no-cpu-baseline --no-side > $O/r4f_cfg3x8_new.json 2> $O/r4f_cfg3x8_new.err || exit 1
(cd build/r03tree && python bench.py --config 3 --seqs-per-gpu 8 --steps 100 --no-cpu-baseline --no-side > $O/r4f_cfg3x8_r03.json 2> $O/r4f_cfg3x8_r03.err) || exit 1
python bench.py --config 5 --steps 100 --no-cpu-baseline --no-side > $O/r4f_cfg5_new.json 2> $O/r4f_cfg5_new.err || exit 1
(cd build/r03tree && python bench.py --config 5 --steps 100 --no-cpu-baseline --no-side > $O/r4f_cfg5_r03.json 2> $O/r4f_cfg5_r03.err) || exit 1
python scripts/summ.py $O/r4f_*.json
