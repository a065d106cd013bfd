#!/bin/bash
# rocprofv3 kernel stats of `bench.py --timed-only` (one launch set at a time, one stream) under an environment setting:
#   bash tools/prof_mode.sh <tag> VAR=value [VAR=value ...]      -> gpurun_out/r03/kstats_<tag>.csv
R=$(pwd); TAG=$1; shift
O=$R/gpurun_out/r03; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof_$TAG
( cd /tmp && env "$@" INSMOS_TWO_STREAMS=0 INSMOS_WINDOWS_IN_FLIGHT=1 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$TAG -o prof --output-format csv -- \
    python $R/bench.py --timed-only --steps 2 --warmup 1 --windows-per-step 16 ) > $O/rocprof_$TAG.log 2>&1
ST=$(find $O/prof_$TAG -name "*kernel_stats.csv" | head -1)
cp "$ST" $O/kstats_$TAG.csv
rm -rf $O/prof_$TAG
