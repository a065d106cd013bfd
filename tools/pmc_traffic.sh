#!/bin/bash
# HBM traffic of the convolution launches from PMC counters: one rocprofv3 pass per counter (--kernel-trace only, never
# combined with other trace domains), over `bench.py --timed-only` with ONE launch set in flight -- every launch of the trace
# belongs to a step.  tools/pmc_traffic.py sums the conv launches, applies the gfx950 FETCH_SIZE correction of
# MI355X_MICROARCH.md and divides by the windows processed.  Usage (GPU box, after one plain bench.py run that cached the
# head calibration):  bash tools/pmc_traffic.sh r02 [windows_per_launch]
R=$(pwd); TAG=${1:-r03}; WPL=${2:-4}; CFG=${3:-cfg2}   # (cfg4: the dense stress scene, launch sets of 2)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_traffic
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_traffic/$c
  ( cd /tmp && INSMOS_TWO_STREAMS=0 INSMOS_WINDOWS_IN_FLIGHT=1 INSMOS_WINDOWS_PER_LAUNCH=$WPL timeout 240 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc_traffic/$c -o p --output-format csv -- \
      python $R/bench.py --timed-only --steps 1 --warmup 1 --windows-per-step $WPL --config $CFG ) > $R/gpurun_out/pmc_traffic/$c.log 2>&1
  echo "$c rc=$?"
done
python $R/tools/pmc_traffic.py $R/gpurun_out/pmc_traffic $TAG $WPL $CFG
