#!/bin/bash
# Round-4 closing measurements on one box: default bench line (with side objects + CPU baseline), the driver's command, the
# other configurations, rocprofv3 kernel stats + PMC traffic per configuration, then profiles/traffic_latest.json.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
python bench.py > $O/r04_bench_line_default.json 2> $O/r04_bench_line_default.err || { tail -5 $O/r04_bench_line_default.err; exit 1; }
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_line_driver.json 2> $O/r04_bench_line_driver.err || exit 1
for c in 2 4 5; do python bench.py --config $c --no-side > $O/r04_bench_line_cfg$c.json 2> $O/r04_bench_line_cfg$c.err || { tail -5 $O/r04_bench_line_cfg$c.err; exit 1; }; done
python bench.py --config 3 --seqs-per-gpu 8 --no-side > $O/r04_bench_line_cfg3x8.json 2> $O/r04_bench_line_cfg3x8.err || exit 1
bash scripts/profile_r04.sh > $O/r04_profile.log 2>&1 || { tail -5 $O/r04_profile.log; exit 1; }
python scripts/make_traffic_latest.py r04 > $O/r04_traffic_latest.log 2>&1; cat $O/r04_traffic_latest.log
cp profiles/traffic_latest.json $O/r04_traffic_latest.json
python - <<'PY'
import json
for t in ("default","driver","cfg2","cfg4","cfg5","cfg3x8"):
    d=json.loads(open(f"gpurun_out/r04_bench_line_{t}.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(t, "us/seq-layer %.2f"%d["selfattn_us_per_layer"], "chain %.3f"%d["chain_frac_of_hbm_peak"], "value %.1f"%d["value"], "roof", r.get("kernel_name"), "frac", r.get("frac"), "launch_us", r.get("launch_us"), "spd", d.get("speedup_vs_dense"), d.get("speedup_vs_batched_dense"))
    for k in ("batched_8seq","cfg5_8seq_gqa"):
        if k in d: print("   ", k, {x: d[k].get(x) for x in ("us_per_sequence_layer","chain_frac_of_hbm_peak","dominant_kernel_launch_us","dominant_kernel_frac_algorithmic","speedup_vs_batched_dense")})
PY
# end to end (random-weight checkpoint in HF layout, Llama-2-7B shapes): one sequence and 8 sequences per step
python scripts/bench_textgen.py --make-checkpoint /tmp/quest_ckpt --generate 64 > $O/r04_e2e_textgen_fused_layers.json 2> $O/r04_e2e_textgen.err || { tail -5 $O/r04_e2e_textgen.err; exit 1; }
python scripts/bench_textgen.py --seqs 8 > $O/r04_e2e_textgen_8seq_fused_layers.json 2> $O/r04_e2e_textgen_8seq.err || { tail -5 $O/r04_e2e_textgen_8seq.err; exit 1; }
cut -c1-700 $O/r04_e2e_textgen_fused_layers.json; cat $O/r04_e2e_textgen_8seq_fused_layers.json
