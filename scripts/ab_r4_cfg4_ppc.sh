#!/bin/bash
# cfg 4 (8191-column rows, 256 selected pages per head): pages per workgroup 16 (default: 16 workgroups per head, 2 per CU)
# vs 32 (8 per head, 1 per CU: half the redundant selections, twice the pages per wave) and 8.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O; R=$O/r04_ab_cfg4_pages_per_workgroup.txt; : > $R
for rep in 1 2; do
for ppc in 0 8 16 32 64; do
    python bench.py --config 4 --pages-per-chunk $ppc --no-side --no-cpu-baseline > $O/ab_ppc.json 2> $O/ab_ppc.err || { tail -5 $O/ab_ppc.err; exit 1; }
    python - "$ppc" >> $R <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_ppc.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
print("cfg 4 pages per workgroup", sys.argv[1], "us/seq-layer %.2f"%d["selfattn_us_per_layer"], "launch", r.get("launch"), "roof_launch_us %.2f"%r.get("launch_us"), "ops", {k: round(v,2) for k,v in (d.get("ops_us") or {}).items() if isinstance(v,(int,float))})
PY
done
done
cat $R
