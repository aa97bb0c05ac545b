#!/bin/bash
# (record of a round-4 experiment: the QUEST_LONG_ROW_WAVES knob and the 16-wave long-row instantiation were removed after this measurement -- 21.3 vs 19.3 us at cfg 4; git log -S QUEST_LONG_ROW_WAVES)
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
QUEST_TUNING=1 QUEST_LONG_ROW_WAVES=16 timeout -k 10 300 python -m pytest tests/test_gpu_long_rows.py -m gpu -q -x -k "8192 or 16384 or 6000" > $O/r4z_tests.log 2>&1; echo "tests exit $?"; tail -3 $O/r4z_tests.log
for ppc in 32 16; do
QUEST_TUNING=1 QUEST_LONG_ROW_WAVES=16 python bench.py --config 4 --steps 200 --no-cpu-baseline --no-side --no-dense --pages-per-chunk $ppc > $O/r4z_cfg4_w16_ppc$ppc.json 2> $O/r4z_cfg4_w16_ppc$ppc.err || { tail -3 $O/r4z_cfg4_w16_ppc$ppc.err; exit 1; }
done
python bench.py --config 4 --steps 200 --no-cpu-baseline --no-side --no-dense > $O/r4z_cfg4_w8.json 2> $O/r4z_cfg4_w8.err || exit 1
python scripts/summ.py $O/r4z_*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4z_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d['roofline']['launch'], d['roofline']['plan'])
PY
