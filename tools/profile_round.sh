#!/bin/bash
# Round evidence on the GPU box: bench line + rocprofv3 kernel stats of the SAME workload (bench.py --timed-only: every launch of
# the trace belongs to a step) with one launch set at a time (sequential: the roofline script's input) and as benchmarked (sets
# overlapping), + the PMC passes.   usage:  bash tools/profile_round.sh r03 [windows_per_launch=8]
# The sequential pass pins everything to ONE stream (INSMOS_TWO_STREAMS=0): the second stream the native runner uses for a single set
# in flight would overlap the kernels whose durations the roofline is computed from.
R=$(pwd); TAG=${1:-r03}; WPL=${2:-8}
O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
python bench.py --steps 10 --warmup 3 --no-extras 2> $O/bench_profile.err | tail -1 > $O/bench_profile.json; cut -c1-300 $O/bench_profile.json
for FL in 1 4; do
  rm -rf $O/prof_fl$FL
  ( cd /tmp && INSMOS_TWO_STREAMS=0 INSMOS_WINDOWS_IN_FLIGHT=$FL INSMOS_WINDOWS_PER_LAUNCH=$WPL timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_fl$FL -o prof --output-format csv -- \
      python $R/bench.py --timed-only --steps 4 --warmup 1 ) > $O/rocprof_fl$FL.log 2>&1
  NWIN=$(grep -o '"windows_total": [0-9]*' $O/rocprof_fl$FL.log | tail -1 | grep -o '[0-9]*$')
  ST=$(find $O/prof_fl$FL -name "*kernel_stats.csv" | head -1)
  cp "$ST" $O/rocprof_kernel_stats_fl$FL.csv
  find $O/prof_fl$FL -name "*kernel_trace.csv" -delete
  python tools/roofline_from_rocprof.py $O/rocprof_kernel_stats_fl$FL.csv --windows ${NWIN:-120} --bench-json $O/bench_profile.json --json > $O/roofline_fl$FL.json
  cat $O/roofline_fl$FL.json
done
head -40 $O/rocprof_kernel_stats_fl1.csv | cut -c1-200
bash tools/pmc_traffic.sh $TAG $WPL | tail -22
bash tools/pmc_mfma.sh $TAG $WPL | tail -30
