#!/bin/bash
O=$(pwd)/gpurun_out/r05g; mkdir -p $O; R=$(pwd)
export TMPDIR=/tmp
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/cfg5 -o p --output-format csv -- python $R/bench.py --config cfg5 --steps 6 --no-cpu-baseline --no-extras ) > $O/cfg5.log 2>&1
tail -1 $O/cfg5.log | cut -c1-300
cp $(find $O/cfg5 -name "*kernel_stats.csv" | head -1) $O/rocprof_kernel_stats_cfg5.csv
find $O/cfg5 -name "*kernel_trace.csv" -delete
head -70 $O/rocprof_kernel_stats_cfg5.csv | cut -c1-180
