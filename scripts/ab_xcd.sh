#!/bin/bash
# usage: scripts/ab_xcd.sh "<bench args>" tag "lib:xcd" ...   lib = base|<variant>, xcd = 0|1 (QUEST_XCD_GROUP)
args="$1"; tag="$2"; shift 2
files=""
for spec in "$@"; do
  lib=${spec%%:*}; x=${spec##*:}
  if [ "$lib" = base ]; then unset QUEST_HIP_LIB; else export QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_$lib.so; fi
  export QUEST_XCD_GROUP=$x
  f=gpurun_out/ab_${tag}_${lib}_xcd${x}.json
  python bench.py $args --no-cpu-baseline --no-dense > $f 2>/dev/null || { echo "FAILED $spec"; exit 1; }
  files="$files $f"
done
python scripts/summ.py $files
