#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_batched.py tests/test_gpu_graph_decode.py -x -q > $O/r3f_tests.log 2>&1 || { tail -30 $O/r3f_tests.log; exit 1; }
tail -2 $O/r3f_tests.log
run() { tag=$1; shift; env $ENVV python bench.py "$@" --steps 300 --no-cpu-baseline --no-dense --no-side > $O/r3f_$tag.json 2> $O/r3f_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3f_$tag.err; }; }
ENVV="QUEST_FE2_PREFILTER=0" run c4_pre0 --config 4
ENVV="QUEST_FE2_PREFILTER=1" run c4_pre1 --config 4
ENVV="QUEST_FE2_PREFILTER=1" run c4_pre1_ppc32 --config 4 --pages-per-chunk 32
ENVV="QUEST_FE2_PREFILTER=2 QUEST_FRONT_END=2" run c3_gen2_pre --config 3
ENVV="QUEST_FE2_PREFILTER=0 QUEST_FRONT_END=2" run c3_gen2_nopre --config 3
ENVV="A=1" run c3_default --config 3
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3f_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0), d['roofline']['plan'])
PY
TL_CONFIG=4 python scripts/timeline.py > $O/r3f_timeline_cfg4.log 2>&1; tail -22 $O/r3f_timeline_cfg4.log
