#!/bin/bash
O=$(pwd)/gpurun_out/r05f; mkdir -p $O; R=$(pwd)
timeout 600 python -m pytest tests/test_gpu_conv.py -q -m gpu -k "wide_masked" --timeout 300 2>&1 | tail -3
export TMPDIR=/tmp
( cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $O/b1 -o b1 --output-format csv -- python $R/tools/b1_trace.py ) > $O/b1_run.log 2>&1
tail -2 $O/b1_run.log
python tools/b1_trace.py --analyze $O/b1 > $O/b1_trace.txt 2>&1; head -100 $O/b1_trace.txt | cut -c1-170
find $O/b1 -name "*kernel_trace.csv" -size +20M -delete
