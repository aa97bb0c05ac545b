#!/bin/bash
# usage: scripts/ab_chain2.sh "<bench args>" tag lib "lead:ppc" ...   (chained launch; run on the GPU box)
args="$1"; tag="$2"; lib="$3"; shift 3
if [ "$lib" = base ]; then unset QUEST_HIP_LIB; else export QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_$lib.so; fi
files=""
for spec in "$@"; do
  export QUEST_CHAIN=1 QUEST_CHAIN_LEAD=${spec%%:*}
  ppc=${spec##*:}
  f=gpurun_out/ab_${tag}_${lib}_lead${QUEST_CHAIN_LEAD}_ppc${ppc}.json
  python bench.py $args --pages-per-chunk $ppc --no-cpu-baseline --no-dense > $f 2>/dev/null || { echo "FAILED $spec"; exit 1; }
  files="$files $f"
done
python scripts/summ.py $files
