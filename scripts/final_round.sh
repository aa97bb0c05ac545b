#!/bin/bash
# Closing measurements of a round on ONE box: the default bench line (side objects + CPU baseline), the driver's command,
# the other configurations, then rocprofv3 kernel stats + PMC traffic per configuration and profiles/traffic_latest.json.
#   scripts/final_round.sh r05 [lines|profiles|e2e]      (default: lines + profiles)
set -o pipefail
R=${1:-r05}; what=${2:-all}
O=$PWD/gpurun_out; mkdir -p $O
if [ "$what" = all ] || [ "$what" = lines ]; then
  python bench.py > $O/${R}_bench_line_default.json 2> $O/${R}_bench_line_default.err || { tail -5 $O/${R}_bench_line_default.err; exit 1; }
  python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${R}_bench_line_driver.json 2> $O/${R}_bench_line_driver.err || exit 1
  for c in 2 4 5; do python bench.py --config $c --no-side > $O/${R}_bench_line_cfg$c.json 2> $O/${R}_bench_line_cfg$c.err || { tail -5 $O/${R}_bench_line_cfg$c.err; exit 1; }; done
  python bench.py --config 3 --seqs-per-gpu 8 --no-side > $O/${R}_bench_line_cfg3x8.json 2> $O/${R}_bench_line_cfg3x8.err || exit 1
fi
if [ "$what" = all ] || [ "$what" = profiles ]; then
  for t in "cfg3:--config 3" "cfg3x8:--config 3 --seqs-per-gpu 8" "cfg5:--config 5" "cfg4:--config 4" "cfg2:--config 2"; do
    tag=${t%%:*}; args=${t#*:}
    bash scripts/profile_round.sh ${R}_$tag $args --steps 20 --warmup 5 --no-side > $O/${R}_profile_$tag.log 2>&1 || { tail -5 $O/${R}_profile_$tag.log; exit 1; }
  done
  python scripts/make_traffic_latest.py $R > $O/${R}_traffic_latest.log 2>&1; cat $O/${R}_traffic_latest.log
  cp profiles/traffic_latest.json $O/${R}_traffic_latest.json
fi
if [ "$what" = e2e ]; then  # end to end (random-weight checkpoint in HF layout, Llama-2-7B shapes): one sequence and 8 per step
  python scripts/bench_textgen.py --make-checkpoint /tmp/quest_ckpt --generate 64 > $O/${R}_e2e_textgen_fused_layers.json 2> $O/${R}_e2e_textgen.err || { tail -5 $O/${R}_e2e_textgen.err; exit 1; }
  python scripts/bench_textgen.py --seqs 8 > $O/${R}_e2e_textgen_8seq_fused_layers.json 2> $O/${R}_e2e_textgen_8seq.err || { tail -5 $O/${R}_e2e_textgen_8seq.err; exit 1; }
  cut -c1-700 $O/${R}_e2e_textgen_fused_layers.json; cat $O/${R}_e2e_textgen_8seq_fused_layers.json
fi
python - "$R" <<'PY'
import json, os, sys
R = sys.argv[1]
for t in ("default", "driver", "cfg2", "cfg4", "cfg5", "cfg3x8"):
    p = f"gpurun_out/{R}_bench_line_{t}.json"
    if not os.path.exists(p):
        continue
    d = json.loads(open(p).read().strip().splitlines()[-1])
    r = d.get("roofline") or {}
    print(t, "us/seq-layer %.2f" % d["selfattn_us_per_layer"], "chain %.3f" % d["chain_frac_of_hbm_peak"], "value %.1f" % d["value"],
          "roof", r.get("kernel_name"), "frac", r.get("frac"), "launch_us", r.get("launch_us"), "spd", d.get("speedup_vs_dense"),
          d.get("speedup_vs_batched_dense"))
    for k in ("batched_8seq", "cfg5_8seq_gqa"):
        if k in d:
            print("   ", k, {x: d[k].get(x) for x in ("us_per_sequence_layer", "chain_frac_of_hbm_peak", "dominant_kernel_launch_us",
                                                      "dominant_kernel_frac_algorithmic", "speedup_vs_batched_dense", "launches_per_layer")})
PY
