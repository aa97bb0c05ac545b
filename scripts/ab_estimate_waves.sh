#!/bin/bash
run() { name=$1; shift; for rep in 1 2; do
  for cfgargs in "cfg3:--config 3 --steps 500" "cfg3x8two:--config 3 --seqs-per-gpu 8 --layer-launches two --steps 200" "cfg3_16k_b1024:--config 3 --seqlen 16384 --token-budget 1024 --steps 300"; do
  tag=${cfgargs%%:*}; args=${cfgargs#*:}
  f=gpurun_out/r06_estw_${name}_${tag}_$rep.json
  "$@" $args --warmup 20 --no-side --no-cpu-baseline > $f 2> ${f%.json}.err
  python - $f "$name $tag #$rep" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); o = d.get("ops_us") or {}
    print(sys.argv[2], "us/seq-layer %.3f chain %.3f | A+E %s us" % (d["selfattn_us_per_layer"], d["chain_frac_of_hbm_peak"], o.get("append_estimate_us") or o.get("two_launch_form_append_estimate_us")), flush=True)
except Exception as e:
    print(sys.argv[2], "FAILED", e, flush=True)
PY
done; done; }
run product python bench.py
run variant_w2 env QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_est_w2.so python bench.py
