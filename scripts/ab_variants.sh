#!/bin/bash
# usage: scripts/ab_variants.sh "<bench args>" tag lib1 lib2 ...   (run on the GPU box; "base" = the default library)
args="$1"; tag="$2"; shift 2
for lib in "$@"; do
  if [ "$lib" = base ]; then unset QUEST_HIP_LIB; else export QUEST_HIP_LIB=$PWD/quest_amd/libquest_hip_$lib.so; fi
  python bench.py $args --no-cpu-baseline --no-dense > gpurun_out/ab_${tag}_${lib}.json 2>/dev/null || { echo "FAILED $lib"; exit 1; }
done
python scripts/summ.py $(for lib in "$@"; do echo gpurun_out/ab_${tag}_${lib}.json; done)
