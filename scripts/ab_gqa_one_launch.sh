#!/bin/bash
# cfg 5 (8 GQA sequences): the layer as one launch with the metadata walks of a group's heads staggered / cached, against the product
run() { name=$1; shift; for rep in 1 2; do
  f=gpurun_out/r06_gqa_${name}_$rep.json
  "$@" > $f 2> ${f%.json}.err
  python - $f "$name #$rep" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d.get("roofline") or {}; o = d.get("ops_us") or {}
    print(sys.argv[2], "us/seq-layer %.3f chain %.3f | %s %.2f us | launches %s" % (d["selfattn_us_per_layer"], d["chain_frac_of_hbm_peak"], r.get("kernel_name"), r.get("launch_us") or 0, d["config"]["launches_per_layer"][:1]), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done; }
B="python bench.py --config 5 --steps 200 --warmup 20 --no-side --no-cpu-baseline"
run two_launches $B --layer-launches two
run one_launch $B --layer-launches one
for v in c q 8 16; do
  run one_launch_variant_$v env QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_gqa_$v.so $B --layer-launches one
done
