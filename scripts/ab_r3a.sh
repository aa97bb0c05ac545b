#!/bin/bash
# Historical (round 3, gpurun_out/r3a_*): needs the round-2 tree as a worktree (`git worktree add --detach build/r02tree <r02 commit>` + `python -m quest_amd.build` there)
# and the in-kernel-merge build of commit 7c41f29 (QUEST_MERGE=inline|launch); kept as the record of how DESIGN.md 3.4's figures were taken.
# Round-3 A/B on one box: r02 tree (build/r02tree) vs this tree with the in-kernel merge and with the merge launch.
set -o pipefail
mkdir -p gpurun_out
O=$PWD/gpurun_out
python -m pytest tests -m gpu -x -q > $O/r3a_tests.log 2>&1 || { tail -30 $O/r3a_tests.log; exit 1; }
tail -2 $O/r3a_tests.log
for cfg in 3 2 4; do
  (cd build/r02tree && python bench.py --config $cfg --steps 300 --no-cpu-baseline > $O/r3a_cfg${cfg}_r02.json 2> $O/r3a_cfg${cfg}_r02.err) || exit 1
  QUEST_MERGE=inline python bench.py --config $cfg --steps 300 --no-cpu-baseline > $O/r3a_cfg${cfg}_inline.json 2> $O/r3a_cfg${cfg}_inline.err || exit 1
  QUEST_MERGE=launch python bench.py --config $cfg --steps 300 --no-cpu-baseline > $O/r3a_cfg${cfg}_launch.json 2> $O/r3a_cfg${cfg}_launch.err || exit 1
  echo "cfg $cfg done"
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r3a_cfg*.json')):
    try: d=json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e: print(f, 'ERR', e); continue
    ops=d.get('ops_us') or {}
    print(f.split('/')[-1], 'us/layer %.2f'%d['selfattn_us_per_layer'], 'AE %.2f'%ops.get('append_estimate_us',0), 'TS+M %.2f'%ops.get('topk_sparse_attn_plus_merge_us',0), 'TS %.2f'%ops.get('topk_sparse_attn_kernel_only_us',0), 'dense', d.get('dense_full_kv_us'), 'spd', d.get('speedup_vs_dense'))
PY
