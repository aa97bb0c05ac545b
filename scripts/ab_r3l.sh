#!/bin/bash
# Needs the round-2 tree as a worktree under build/r02tree (see ab_r3a.sh).
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
for rep in 1 2; do
(cd build/r02tree && python bench.py --config 2 --steps 300 --no-cpu-baseline > $O/r3l_cfg2_r02_$rep.json 2> $O/r3l_cfg2_r02_$rep.err) || exit 1
python bench.py --config 2 --steps 300 --no-cpu-baseline --no-side > $O/r3l_cfg2_new_$rep.json 2> $O/r3l_cfg2_new_$rep.err || exit 1
done
(cd build/r02tree && python bench.py --config 3 --steps 300 --no-cpu-baseline > $O/r3l_cfg3_r02.json 2> $O/r3l_cfg3_r02.err) || exit 1
python bench.py --config 3 --steps 300 --no-cpu-baseline --no-side > $O/r3l_cfg3_new.json 2> $O/r3l_cfg3_new.err || exit 1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3l_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'dense', d.get('dense_full_kv_us'), (d.get('reference_op_sequence_us') or {}).get('append_us'))
PY
