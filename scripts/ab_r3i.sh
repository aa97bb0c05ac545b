#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
run() { tag=$1; shift; env $ENVV python bench.py "$@" --steps 300 --no-cpu-baseline --no-dense --no-side > $O/r3i_$tag.json 2> $O/r3i_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3i_$tag.err; }; }
ENVV="QUEST_FE_SPECIALIZE=0" run c3_generic --config 3
ENVV="QUEST_FE_SPECIALIZE=1" run c3_special --config 3
ENVV="QUEST_FE_SPECIALIZE=0" run c3_generic2 --config 3
ENVV="QUEST_FE_SPECIALIZE=1" run c3_special2 --config 3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3i_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0))
PY
