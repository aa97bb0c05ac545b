#!/bin/bash
# Round-3: pages per workgroup at cfg 3 / cfg 4 (one workgroup per CU vs two), and fresh in-kernel timelines.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
run() { tag=$1; shift; python bench.py "$@" --steps 300 --no-cpu-baseline --no-dense --no-side > $O/r3e_$tag.json 2> $O/r3e_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3e_$tag.err; }; }
run c3_ppc0 --config 3
run c3_ppc16 --config 3 --pages-per-chunk 16
run c3_ppc0b --config 3
run c4_ppc0 --config 4
run c4_ppc32 --config 4 --pages-per-chunk 32
run c4_ppc22 --config 4 --pages-per-chunk 22
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3e_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0), d['roofline']['plan'])
PY
TL_CONFIG=3 python scripts/timeline.py > $O/r3e_timeline_cfg3.log 2>&1; tail -45 $O/r3e_timeline_cfg3.log
TL_CONFIG=4 python scripts/timeline.py > $O/r3e_timeline_cfg4.log 2>&1; tail -22 $O/r3e_timeline_cfg4.log
