#!/bin/bash
# A/B of the batched layer forms on one box: one launch per layer (csrc/layer_device.cuh) vs the two launches.
# usage: scripts/ab_layer_launches.sh <tag> [configs: "3x8 5"]
set -e
tag=${1:-ab}; cfgs=${2:-"3x8 5"}
mkdir -p gpurun_out
for c in $cfgs; do
  for form in two one; do
    if [ "$c" = "3x8" ]; then args="--config 3 --seqs-per-gpu 8"; else args="--config $c"; fi
    timeout -k 10 400 python bench.py $args --steps 100 --no-cpu-baseline --no-side --layer-launches $form \
      > gpurun_out/${tag}_cfg${c}_${form}.json 2> gpurun_out/${tag}_cfg${c}_${form}.err
    python - <<PY
import json
d = json.load(open("gpurun_out/${tag}_cfg${c}_${form}.json"))
r, o = d.get("roofline") or {}, d.get("ops_us") or {}
print("cfg${c} ${form}: us/seq-layer %.3f chain %.3f kernel %s %.2f us frac %.3f vs batched dense %s ops %s" % (
    d["selfattn_us_per_layer"], d["chain_frac_of_hbm_peak"], r.get("kernel_name"), r.get("launch_us") or 0, r.get("frac") or 0,
    d.get("speedup_vs_batched_dense"), {k: round(v, 2) for k, v in o.items() if isinstance(v, float)}))
PY
  done
done
