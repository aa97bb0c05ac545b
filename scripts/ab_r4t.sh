#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_long_rows.py tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_batched.py -m gpu -q -x > $O/r4t_tests.log 2>&1; echo "tests exit $?"; tail -3 $O/r4t_tests.log
for rep in 1 2; do
python bench.py --config 4 --steps 200 --no-cpu-baseline --no-side --no-dense > $O/r4t_cfg4_new_$rep.json 2> $O/r4t_cfg4_new_$rep.err || exit 1
(cd build/r03tree && python bench.py --config 4 --steps 200 --no-cpu-baseline --no-side --no-dense > $O/r4t_cfg4_r03_$rep.json 2> $O/r4t_cfg4_r03_$rep.err) || exit 1
done
python scripts/summ.py $O/r4t_*.json
