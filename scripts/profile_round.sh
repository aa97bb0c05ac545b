#!/bin/bash
# Run on the GPU box: rocprofv3 kernel-trace stats + PMC traffic of bench.py for one configuration.
#   scripts/profile_round.sh <tag> <bench args...>
# Outputs under gpurun_out/prof_<tag>/: stats/ (kernel_stats.csv), fetch/, write/ (counter_collection.csv), line.json
set -e
tag=$1; shift
out=gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o r -- python3 bench.py "$@" > $out/line.json 2> $out/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o r -- python3 bench.py "$@" --no-cpu-baseline --no-dense > /dev/null 2> $out/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o r -- python3 bench.py "$@" --no-cpu-baseline --no-dense > /dev/null 2> $out/write.err
python3 scripts/pmc_summary.py $out/pmc_traffic.json $out/fetch $out/write > $out/pmc_summary.txt
# keep only the summaries (the traces are large)
f=$(find $out/stats -name "*kernel_stats.csv" | head -1); cp $f $out/kernel_stats.csv
rm -rf $out/stats $out/fetch $out/write
ls -la $out
