#!/bin/bash
# round 6: rocprofv3 kernel stats of the cfg-5 training step (bench.py --config cfg5) -> gpurun_out/<tag>/cfg5_kernel_stats.csv
R=$(pwd); TAG=${1:-r06_cfg5}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
timeout 300 python bench.py --config cfg5 --steps 8 2> $O/bench_cfg5.err | tail -1 > $O/bench_cfg5.json; cut -c1-400 $O/bench_cfg5.json
timeout 300 python bench.py --config cfg5 --steps 8 --train-bf16 2> $O/bench_cfg5_bf16.err | tail -1 > $O/bench_cfg5_bf16.json; cut -c1-400 $O/bench_cfg5_bf16.json
rm -rf $O/prof
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof -o prof --output-format csv -- python $R/bench.py --config cfg5 --steps 8 --no-cpu-baseline ) > $O/rocprof.log 2>&1
ST=$(find $O/prof -name "*kernel_stats.csv" | head -1); cp "$ST" $O/cfg5_kernel_stats.csv
find $O/prof -name "*kernel_trace.csv" -delete
head -3 $O/cfg5_kernel_stats.csv | cut -c1-200
