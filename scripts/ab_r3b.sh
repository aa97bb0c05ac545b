#!/bin/bash
# Round-3: runtime knobs of the HIP runtime on the headline step (cfg 3, graph replay), one box.
set -o pipefail
mkdir -p gpurun_out
O=$PWD/gpurun_out
run() { tag=$1; shift; env "$@" python bench.py --config 3 --steps 300 --no-cpu-baseline --no-dense > $O/r3b_$tag.json 2> $O/r3b_$tag.err || { echo "FAILED $tag"; tail -3 $O/r3b_$tag.err; }; }
run base A=1
run devkernarg0 HIP_FORCE_DEV_KERNARG=0
run devkernarg1 HIP_FORCE_DEV_KERNARG=1
run pktcap0 DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run pktcap1 DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run hwq1 GPU_MAX_HW_QUEUES=1
run base2 A=1
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3b_*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0))
PY
