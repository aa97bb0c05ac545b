#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_graph_decode.py tests/test_gpu_full_size.py tests/test_gpu_estimate_semantics.py -x -q > $O/r3s_tests.log 2>&1 || { tail -30 $O/r3s_tests.log; exit 1; }
tail -2 $O/r3s_tests.log
run() { tag=$1; shift; env $ENVV timeout -k 10 200 python bench.py "$@" --steps 300 --no-cpu-baseline --no-dense --no-side > $O/r3s_$tag.json 2> $O/r3s_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3s_$tag.err; exit 1; }; }
for rep in 1 2; do
ENVV="QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_nolowbins.so" run c3_range$rep --config 3
ENVV="A=1" run c3_lowbins$rep --config 3
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3s_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1])
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0))
PY
TL_CONFIG=3 timeout -k 10 200 python scripts/timeline.py > $O/r3s_timeline_cfg3.log 2>&1; head -22 $O/r3s_timeline_cfg3.log
