#!/bin/bash
# Round-3 closing measurements on one box: default bench line (with side objects + CPU baseline), the other configurations,
# rocprofv3 kernel stats + PMC traffic per configuration, kernel sweep with its own kernel stats.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
python bench.py > $O/r03_bench_line_default.json 2> $O/r03_bench_line_default.err || exit 1
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r03_bench_line_driver.json 2> $O/r03_bench_line_driver.err || exit 1
for c in 2 4 5; do python bench.py --config $c --no-side > $O/r03_bench_line_cfg$c.json 2> $O/r03_bench_line_cfg$c.err || exit 1; done
python bench.py --config 3 --seqs-per-gpu 8 --no-side > $O/r03_bench_line_cfg3x8.json 2> $O/r03_bench_line_cfg3x8.err || exit 1
bash scripts/profile_r03.sh > $O/r03_profile.log 2>&1 || { tail -5 $O/r03_profile.log; exit 1; }
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_r03_sweep -o r -- python3 bench.py --kernel-sweep > $O/r03_sweep.log 2> $O/r03_sweep.md || { tail -5 $O/r03_sweep.md; exit 1; }
f=$(find $O/prof_r03_sweep -name "*kernel_stats.csv" | head -1); cp $f $O/r03_kernel_sweep_kernel_stats.csv; rm -rf $O/prof_r03_sweep
python - <<'PY'
import json
for t in ("default","driver","cfg2","cfg4","cfg5","cfg3x8"):
    d=json.loads(open(f"gpurun_out/r03_bench_line_{t}.json").read().strip().splitlines()[-1])
    r=d.get("roofline") or {}
    print(t, "us/seq-layer %.2f"%d["selfattn_us_per_layer"], "chain %.3f"%d["chain_frac_of_hbm_peak"], "value %.1f"%d["value"], "roof frac", r.get("frac"), "launch_us", r.get("launch_us"), "spd", d.get("speedup_vs_dense"), d.get("speedup_vs_batched_dense"))
PY
