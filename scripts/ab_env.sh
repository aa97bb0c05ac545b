#!/bin/bash
# A/B of tuning environment variables on one box (QUEST_TUNING=1 is set): each "NAME=VAL[,NAME=VAL]" set runs the bench once.
# usage: scripts/ab_env.sh <tag> "<bench args>" <set> [<set> ...]      (set "-" = no variable)
tag=$1; args=$2; shift 2
export QUEST_TUNING=1
for set in "$@"; do
  name=$(echo "$set" | tr ',=/' '___' | tail -c 60)
  ( if [ "$set" != "-" ]; then for kv in $(echo "$set" | tr ',' ' '); do export "$kv"; done; fi
    timeout -k 10 400 python bench.py $args --no-cpu-baseline --no-side > gpurun_out/${tag}_${name}.json 2> gpurun_out/${tag}_${name}.err )
  python - <<PY
import json
try:
    d = json.load(open("gpurun_out/${tag}_${name}.json"))
    r, o = d.get("roofline") or {}, d.get("ops_us") or {}
    print("${set}: us/seq-layer %.3f chain %.3f kernel %s %.2f us frac %.3f vs dense %s / batched %s" % (
        d["selfattn_us_per_layer"], d["chain_frac_of_hbm_peak"], r.get("kernel_name"), r.get("launch_us") or 0, r.get("frac") or 0,
        d.get("speedup_vs_dense"), d.get("speedup_vs_batched_dense")))
except Exception as e:
    print("${set}: FAILED", e)
PY
done
