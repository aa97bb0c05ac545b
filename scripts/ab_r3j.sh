#!/bin/bash
# Historical (round 3, gpurun_out/r3j_*): libquest_hip_nohints.so = `build_variant(..., ["-DQUEST_NO_LAYOUT_HINTS"])` of the tree that still had hints in the attention kernel.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/r3j_tests.log 2>&1 || { tail -30 $O/r3j_tests.log; exit 1; }
tail -2 $O/r3j_tests.log
run() { tag=$1; shift; env $ENVV python bench.py "$@" --steps 300 --no-cpu-baseline --no-dense --no-side > $O/r3j_$tag.json 2> $O/r3j_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3j_$tag.err; }; }
for rep in 1 2; do
ENVV="QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_nohints.so" run c3_nohints$rep --config 3
ENVV="A=1" run c3_hints$rep --config 3
done
ENVV="QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_nohints.so" run c4_nohints --config 4
ENVV="A=1" run c4_hints --config 4
ENVV="QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_nohints.so" run c3x8_nohints --config 3 --seqs-per-gpu 8
ENVV="A=1" run c3x8_hints --config 3 --seqs-per-gpu 8
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3j_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0))
PY
