#!/bin/bash
# usage: scripts/ab_direct.sh "<bench args>" tag   -- staged (QUEST_FE1_DIRECT=0) vs direct register ownership, twice each
args="$1"; tag="$2"
files=""
for rep in 1 2; do for d in 0 1; do
  export QUEST_FE1_DIRECT=$d
  f=gpurun_out/ab_${tag}_direct${d}_r${rep}.json
  python bench.py $args --no-cpu-baseline --no-dense > $f 2>/dev/null || { echo "FAILED $d"; exit 1; }
  files="$files $f"
done; done
python scripts/summ.py $files
