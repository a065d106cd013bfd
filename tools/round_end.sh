#!/bin/bash
# Round-end evidence on the GPU box:  bash tools/round_end.sh r04 [suite|profile|all]  -> gpurun_out/<tag>/ (copy what is to be judged
# into profiles/).  Two halves so that one gpurun call stays short: `suite` = whole GPU suite + smoke + the bench line;
# `profile` = rocprofv3 kernel stats + PMC passes of the same workload, per-layer table, cfg-4, cfg-5, latency, driver.
R=$(pwd); TAG=${1:-r05}; WHAT=${2:-all}; O=$R/gpurun_out/$TAG; mkdir -p $O
if [ "$WHAT" = suite ] || [ "$WHAT" = all ]; then
  echo "== whole GPU suite"
  timeout 1200 python -m pytest tests -m gpu -q --timeout 400 > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-300
  echo "== smoke"
  timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
  echo "== bench"
  timeout 400 python bench.py --steps 20 --warmup 3 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-300 $O/bench.json
fi
if [ "$WHAT" = profile ] || [ "$WHAT" = all ]; then
  echo "== profile round"
  bash tools/profile_round.sh $TAG 8 > $O/profile_round.log 2>&1; head -3 $O/profile_round.log | cut -c1-400
  echo "== layers"
  timeout 300 python tools/batch_layers.py 8 $O/layers_b8.csv 2>&1 | grep -v amdgpu.ids | tail -1
  echo "== cfg4"
  timeout 600 python bench.py --config cfg4 2> $O/bench_cfg4.err | tail -1 > $O/bench_cfg4.json; cut -c1-200 $O/bench_cfg4.json
  echo "== cfg5 (training step)"
  timeout 400 python bench.py --config cfg5 --steps 8 2> $O/bench_cfg5.err | tail -1 > $O/bench_cfg5.json; cut -c1-200 $O/bench_cfg5.json
  echo "== latency"
  timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency_probe.txt; tail -3 $O/latency_probe.txt
  echo "== driver (I/O inclusive)"
  INSMOS_BENCH_DIR=/dev/shm timeout 300 python tools/driver_bench.py 480 2>&1 | grep -v amdgpu.ids > $O/driver_bench.txt; grep "scans/s" $O/driver_bench.txt
fi
