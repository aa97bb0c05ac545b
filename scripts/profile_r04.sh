#!/bin/bash
# Round-4 profiles: rocprofv3 kernel stats + PMC traffic of bench.py per configuration (gpurun_out/prof_r04_<tag>/).
set -o pipefail
for spec in "cfg3:--config 3" "cfg3x8:--config 3 --seqs-per-gpu 8" "cfg5:--config 5" "cfg4:--config 4" "cfg2:--config 2"; do
  tag=${spec%%:*}; args=${spec#*:}
  bash scripts/profile_round.sh r04_$tag $args --steps 20 --warmup 5 --no-side || { echo "FAILED $tag"; exit 1; }
  echo "== $tag"; head -8 gpurun_out/prof_r04_$tag/kernel_stats.csv | cut -c1-160
done
