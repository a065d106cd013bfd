#!/bin/bash
# per-kernel durations of the segmented BatchNorm at a few layer shapes (rocprofv3 kernel stats) -> gpurun_out/<tag>/bn_kernels.txt
R=$(pwd); TAG=${1:-r06_bn}; O=$R/gpurun_out/$TAG; mkdir -p $O; export TMPDIR=/tmp
: > $O/bn_kernels.txt
for shape in "1570000 8" "689000 16" "261000 32" "200000 64" "108000 128" "300000 256"; do
  rm -rf $O/prof
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/tools/bn_kernel_probe.py $shape ) 2>&1 | grep "^rows" >> $O/bn_kernels.txt
  ST=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  python - "$ST" >> $O/bn_kernels.txt <<'PY'
import csv, sys, re
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_bnseg_\w+(<\w+>)?)", r["Name"])
    if m:
        print("   %-32s calls %4s avg_us %7.1f" % (m.group(1), r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
cat $O/bn_kernels.txt
