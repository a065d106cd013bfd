#!/bin/bash
# round-end evidence: GPU tests, smoke, bench JSON and the rocprofv3 kernel-trace stats of the same bench command
R=$(pwd); TAG=${1:-r01}
mkdir -p $R/gpurun_out/$TAG
python -m pytest tests -m gpu -q 2>&1 | tail -5 > $R/gpurun_out/$TAG/pytest_gpu.log; cat $R/gpurun_out/$TAG/pytest_gpu.log
python __graft_entry__.py --smoke 2>&1 | tail -2 > $R/gpurun_out/$TAG/smoke.log; cat $R/gpurun_out/$TAG/smoke.log
python bench.py 2>&1 | tail -1 > $R/gpurun_out/$TAG/bench.json; cut -c1-400 $R/gpurun_out/$TAG/bench.json
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/$TAG/prof -o prof --output-format csv -- python $R/bench.py --steps 5 --no-cpu-baseline > $R/gpurun_out/$TAG/rocprof_bench.log 2>&1
ls $R/gpurun_out/$TAG/prof | head; rm -f $R/gpurun_out/$TAG/prof/*kernel_trace.csv
