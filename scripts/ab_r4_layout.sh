#!/bin/bash
# A/B of the KV / metadata pool layout (NHD = the reference's default, HND = per-head contiguous pages) on one box.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O; R=$O/r04_ab_layout.txt; : > $R
for rep in 1 2; do
for lay in NHD HND; do
  for spec in "3 1" "3 8" "4 1" "5 8" "2 1"; do
    set -- $spec
    python bench.py --config $1 --seqs-per-gpu $2 --layout $lay --no-side --no-cpu-baseline > $O/ab_layout.json 2> $O/ab_layout.err || { tail -5 $O/ab_layout.err; exit 1; }
    python - "$lay" "$1" "$2" >> $R <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_layout.json").read().strip().splitlines()[-1]); r=d.get("roofline") or {}
print(sys.argv[1], "cfg", sys.argv[2], "seqs", sys.argv[3], "us/seq-layer %.2f"%d["selfattn_us_per_layer"], "chain %.3f"%d["chain_frac_of_hbm_peak"], "kernel", r.get("launch"), "launch_us", r.get("launch_us"), "frac", r.get("frac"))
PY
  done
done
done
cat $R
