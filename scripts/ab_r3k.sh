#!/bin/bash
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
python -m pytest tests/test_gpu_full_size.py tests/test_gpu_fuzz.py tests/test_gpu_batched.py tests/test_gpu_graph_decode.py tests/test_gpu_parity.py -x -q > $O/r3k_tests.log 2>&1 || { tail -30 $O/r3k_tests.log; exit 1; }
tail -2 $O/r3k_tests.log
run() { tag=$1; shift; env $ENVV python bench.py "$@" --steps 300 --no-cpu-baseline --no-dense --no-side > $O/r3k_$tag.json 2> $O/r3k_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3k_$tag.err; }; }
for c in "c3:--config 3" "c3x8:--config 3 --seqs-per-gpu 8" "c5:--config 5" "c4:--config 4"; do
  t=${c%%:*}; a=${c#*:}
  ENVV="QUEST_FE_SPECIALIZE=0" run ${t}_generic $a
  ENVV="QUEST_FE_SPECIALIZE=1" run ${t}_special $a
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3k_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0))
PY
