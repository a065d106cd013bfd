#!/bin/bash
# Round-end evidence on the GPU box:  bash tools/round_end.sh r03   -> gpurun_out/<tag>/ (copy what is to be judged into profiles/)
R=$(pwd); TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
echo "== whole GPU suite"
timeout 1800 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -3 $O/pytest_gpu.log | cut -c1-300
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log
echo "== profile round"
bash tools/profile_round.sh $TAG 8 > $O/profile_round.log 2>&1; head -3 $O/profile_round.log | cut -c1-400
echo "== layers"
timeout 300 python tools/batch_layers.py 8 $O/layers_b8.csv 2>&1 | grep -v amdgpu.ids | tail -1
echo "== cfg4"
timeout 600 python bench.py --config cfg4 2> $O/bench_cfg4.err | tail -1 > $O/bench_cfg4.json; cut -c1-200 $O/bench_cfg4.json
echo "== latency"
timeout 300 python tools/latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/latency_probe.txt; tail -3 $O/latency_probe.txt
echo "== training step"
timeout 300 python tools/train_step_bench.py 1886 1 2>&1 | grep -v amdgpu.ids > $O/train_step_b1.log; grep "full training step" $O/train_step_b1.log
timeout 300 python tools/train_step_bench.py 1886 4 2>&1 | grep -v amdgpu.ids > $O/train_step_b4.log; grep "full training step" $O/train_step_b4.log
