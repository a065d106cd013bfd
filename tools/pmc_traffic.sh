#!/bin/bash
# HBM traffic of the sparse-conv launches from PMC counters: one rocprofv3 pass per counter (--kernel-trace only), on a
# single window (one in flight) -- tools/pmc_traffic.py sums the last window's conv launches and applies the gfx950
# FETCH_SIZE correction of MI355X_MICROARCH.md.  Usage (GPU box): bash tools/pmc_traffic.sh r01
R=$(pwd); TAG=${1:-r01}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && INSMOS_WINDOWS_IN_FLIGHT=1 timeout 170 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_traffic/$c -o p --output-format csv -- \
      python $R/bench.py --steps 1 --warmup 1 --windows-per-step 1 --no-cpu-baseline ) > $R/gpurun_out/pmc_traffic/$c.log 2>&1
  echo "$c rc=$?"
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_traffic $TAG
