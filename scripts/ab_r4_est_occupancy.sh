#!/bin/bash
# GQA estimate (G = 4: 71 VGPRs -> 7 workgroups per CU resident of the 8 a cfg-5 launch gives each CU): the same kernel
# built for 8 waves per SIMD (64 VGPRs, 8 bytes of scratch) and with 2 load rounds per wave instead of 4.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O; R=$O/r04_ab_estimate_occupancy.txt; : > $R
python - <<'PY'
from quest_amd.build import build_variant
build_variant("quest_amd/libquest_hip_est8.so", ["-DQUEST_EST_MIN_WAVES=8"])
build_variant("quest_amd/libquest_hip_estiter2.so", ["-DQUEST_EST_ITER_GQA=2"])
PY
for rep in 1 2; do
for v in "" est8 estiter2; do
  for spec in "5 8" "4 1"; do
    set -- $spec
    if [ -n "$v" ]; then export QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_$v.so; else unset QUEST_HIP_LIB; fi
    python bench.py --config $1 --seqs-per-gpu $2 --no-side --no-cpu-baseline > $O/ab_est.json 2> $O/ab_est.err || { tail -5 $O/ab_est.err; exit 1; }
    python - "${v:-default}" "$1" >> $R <<'PY'
import json,sys
d=json.loads(open("gpurun_out/ab_est.json").read().strip().splitlines()[-1]); o=d.get("ops_us") or {}
print(sys.argv[1], "cfg", sys.argv[2], "us/seq-layer %.2f"%d["selfattn_us_per_layer"], "A+E us %.2f"%o.get("append_estimate_us", 0), "frac %.3f"%o.get("append_estimate_frac_of_hbm_peak", 0))
PY
  done
done
done
cat $R
