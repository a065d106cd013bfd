#!/bin/bash
# round 2, GPU call 3: batched parity tests, the whole GPU suite, launch-set sweep, per-layer table of a launch set, bench
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== batched parity tests"
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_zz_gpu_batched_motionnet.py -q 2>&1 | tail -25 | cut -c1-300
echo "== whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -15 | cut -c1-300
echo "== sweep"
timeout 400 python tools/batch_sweep.py 16 6 2>&1 | grep -v amdgpu.ids | tail -14
INSMOS_SPLIT_TAP_MOD=0 timeout 300 python tools/batch_sweep.py 16 6 2>&1 | grep -v amdgpu.ids | tail -12
echo "== layers B=4 / B=8"
timeout 300 python tools/batch_layers.py 4 $R/gpurun_out/r02/layers_b4.csv 2>&1 | grep -v amdgpu.ids | tail -75
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8.csv 2>&1 | grep -v amdgpu.ids | tail -3
timeout 300 python tools/batch_layers.py 1 $R/gpurun_out/r02/layers_b1.csv 2>&1 | grep -v amdgpu.ids | tail -3
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_b.json 2> $R/gpurun_out/r02/bench_b.err; tail -c 3000 $R/gpurun_out/r02/bench_b.json; tail -3 $R/gpurun_out/r02/bench_b.err
