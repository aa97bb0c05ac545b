#!/bin/bash
# (record: QUEST_FUSED_WAVES=16 existed only in the experiment tree -- 16-wave workgroups at cfg 3, 13.3-15.1 vs 12.07 us; git log -S QUEST_FUSED_WAVES.  QUEST_COLRANGE=1 with QUEST_TUNING=1 still selects column ranges.)
# Round 4: 8-wave vs 16-wave workgroups (one per CU), column-range vs slot ownership, cfg 3.
set -o pipefail
O=$PWD/gpurun_out; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" python bench.py --config 3 --steps 300 --no-cpu-baseline --no-side > $O/r4d_$name.json 2> $O/r4d_$name.err || { tail -3 $O/r4d_$name.err; return 1; }
}
for rep in 1 2; do
run cr8_$rep QUEST_TUNING=1 || exit 1
run sl8_$rep QUEST_TUNING=1 QUEST_COLRANGE=0 || exit 1
run cr16_$rep QUEST_TUNING=1 QUEST_FUSED_WAVES=16 || exit 1
run sl16_$rep QUEST_TUNING=1 QUEST_FUSED_WAVES=16 QUEST_COLRANGE=0 || exit 1
done
python scripts/summ.py $O/r4d_*.json
