#!/bin/bash
# A/B of the pool layouts on one box: every configuration once per layout, interleaved (A B A B) so that box drift shows.
# usage: scripts/ab_layout.sh <tag> [layouts...]      default layouts: NHD NHD_ROT
tag=${1:-r06_ab_layout}; shift
layouts=${@:-NHD NHD_ROT}
run() {  # name, bench args
  for rep in 1 2; do for lay in $layouts; do
    f=gpurun_out/${tag}_$1_${lay}_$rep.json
    timeout -k 10 400 python bench.py $2 --layout $lay --no-cpu-baseline --no-side > $f 2> ${f%.json}.err
    python - <<PY
import json
try:
    d = json.load(open("$f"))
    r, o = d.get("roofline") or {}, d.get("ops_us") or {}
    print("$1 $lay #$rep: us/seq-layer %.3f chain %.3f | %s %.2f us frac %.3f | A+E %s T+S %s | vs dense %s / batched %s" % (
        d["selfattn_us_per_layer"], d["chain_frac_of_hbm_peak"], r.get("kernel_name"), r.get("launch_us") or 0, r.get("frac") or 0,
        o.get("append_estimate_us") or o.get("two_launch_form_append_estimate_us"), o.get("topk_sparse_attn_us") or o.get("two_launch_form_topk_sparse_attn_us"),
        d.get("speedup_vs_dense"), d.get("speedup_vs_batched_dense")), flush=True)
except Exception as e:
    print("$1 $lay #$rep: FAILED", e, flush=True)
PY
  done; done
}
run cfg3x8 "--config 3 --seqs-per-gpu 8 --steps 200 --warmup 20"
run cfg5 "--config 5 --steps 200 --warmup 20"
run cfg3 "--config 3 --steps 500 --warmup 20"
run cfg4 "--config 4 --steps 300 --warmup 20"
run cfg2 "--config 2 --steps 300 --warmup 20"
