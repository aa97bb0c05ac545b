#!/bin/bash
# (record of a round-4 experiment: the QUEST_SHARED_WAVES knob and the 8-wave group-shared instantiations were removed after this measurement -- 18.4 vs 17.8 us at cfg 2; git log -S QUEST_SHARED_WAVES)
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for rep in 1 2; do
python bench.py --config 2 --steps 300 --no-cpu-baseline --no-side > $O/r4j_cfg2_w8_$rep.json 2> $O/r4j_cfg2_w8_$rep.err || exit 1
QUEST_TUNING=1 QUEST_SHARED_WAVES=4 python bench.py --config 2 --steps 300 --no-cpu-baseline --no-side > $O/r4j_cfg2_w4_$rep.json 2> $O/r4j_cfg2_w4_$rep.err || exit 1
done
python bench.py --config 3 --steps 300 --no-cpu-baseline --no-side > $O/r4j_cfg3_w8.json 2> $O/r4j_cfg3_w8.err || exit 1
QUEST_TUNING=1 QUEST_SHARED_WAVES=4 python bench.py --config 3 --steps 300 --no-cpu-baseline --no-side > $O/r4j_cfg3_w4.json 2> $O/r4j_cfg3_w4.err || exit 1
python scripts/summ.py $O/r4j_*.json
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r4j_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], 'dense_us', d.get('dense_full_kv_us'), 'dense GB/s', d.get('dense_gbs'), (d.get('roofline') or {}).get('launch_us'))
PY
