#!/bin/bash
# Round evidence on the GPU box: bench line + rocprofv3 kernel stats of the SAME workload (bench.py --timed-only: every launch of
# the trace belongs to a step) with one launch set at a time (sequential: the roofline script's input) and as benchmarked (sets
# overlapping), + the PMC passes.   usage:  bash tools/profile_round.sh r02 [windows_per_launch=8]
R=$(pwd); TAG=${1:-r02}; WPL=${2:-8}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
for FL in 1 3; do
  rm -rf $O/prof_fl$FL
  ( cd /tmp && INSMOS_WINDOWS_IN_FLIGHT=$FL INSMOS_WINDOWS_PER_LAUNCH=$WPL timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fl$FL -o prof --output-format csv -- \
      python $R/bench.py --timed-only --steps 4 --warmup 1 ) > $O/rocprof_fl$FL.log 2>&1
  ST=$(find $O/prof_fl$FL -name "*kernel_stats.csv" | head -1)
  cp "$ST" $O/rocprof_kernel_stats_fl$FL.csv
  find $O/prof_fl$FL -name "*kernel_trace.csv" -delete
  python tools/roofline_from_rocprof.py $O/rocprof_kernel_stats_fl$FL.csv --windows 120 --bench-json $O/bench.json --json > $O/roofline_fl$FL.json
  cat $O/roofline_fl$FL.json
done
head -40 $O/rocprof_kernel_stats_fl1.csv | cut -c1-200
bash tools/pmc_traffic.sh $TAG $WPL | tail -22
bash tools/pmc_mfma.sh $TAG $WPL | tail -30
