#!/bin/bash
# usage: scripts/ab_env.sh "<bench args>" tag "lib:FE" ...   lib = base|<variant>, FE = 0|1|2 (QUEST_FRONT_END)
args="$1"; tag="$2"; shift 2
files=""
for spec in "$@"; do
  lib=${spec%%:*}; fe=${spec##*:}
  if [ "$lib" = base ]; then unset QUEST_HIP_LIB; else export QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_$lib.so; fi
  export QUEST_FRONT_END=$fe
  f=gpurun_out/ab_${tag}_${lib}_fe${fe}.json
  python bench.py $args --no-cpu-baseline --no-dense > $f 2>/dev/null || { echo "FAILED $spec"; exit 1; }
  files="$files $f"
done
python scripts/summ.py $files
