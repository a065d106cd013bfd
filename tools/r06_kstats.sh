#!/bin/bash
# round 6: rocprofv3 kernel stats of bench.py --timed-only, ONE launch set at a time on one stream -> gpurun_out/<tag>/kernel_stats.csv
R=$(pwd); TAG=${1:-r06_kstats}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
rm -rf $O/prof; [ -f /tmp/insmos_bench_calibration.json ] || python tools/calibrate.py 2>&1 | tail -1
( cd /tmp && INSMOS_TWO_STREAMS=0 INSMOS_WINDOWS_IN_FLIGHT=1 INSMOS_WINDOWS_PER_LAUNCH=8 timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o prof --output-format csv -- \
    python $R/bench.py --timed-only --steps 4 --warmup 1 ) > $O/rocprof.log 2>&1
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$ST" $O/kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
grep -o '"windows_total": [0-9]*' $O/rocprof.log | tail -1
head -5 $O/kernel_stats.csv | cut -c1-150
