#!/bin/bash
# GQA estimate (cfg 5's A+E launch; cfg 4's): tile-shape tuning builds against the product, op time + whole step
run() { name=$1; shift; for rep in 1 2; do for cfg in 5 4; do
  f=gpurun_out/r06_est_${name}_cfg${cfg}_$rep.json
  "$@" --config $cfg --steps 200 --warmup 20 --no-side --no-cpu-baseline > $f 2> ${f%.json}.err
  python - $f "$name cfg $cfg #$rep" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); o = d.get("ops_us") or {}
    print(sys.argv[2], "us/seq-layer %.3f chain %.3f | A+E %.2f us" % (d["selfattn_us_per_layer"], d["chain_frac_of_hbm_peak"], o.get("append_estimate_us") or -1), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done; done; }
run product python bench.py
for v in i8 w8 w2; do run variant_$v env QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_est_$v.so python bench.py; done
