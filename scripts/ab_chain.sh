#!/bin/bash
# usage: scripts/ab_chain.sh "<bench args>" tag "chain:lead" ...   (QUEST_CHAIN=0|1, QUEST_CHAIN_LEAD; run on the GPU box)
args="$1"; tag="$2"; shift 2
files=""
for spec in "$@"; do
  export QUEST_CHAIN=${spec%%:*} QUEST_CHAIN_LEAD=${spec##*:}
  f=gpurun_out/ab_${tag}_chain${QUEST_CHAIN}_lead${QUEST_CHAIN_LEAD}.json
  python bench.py $args --no-cpu-baseline --no-dense > $f 2>/dev/null || { echo "FAILED $spec"; exit 1; }
  files="$files $f"
done
python scripts/summ.py $files
