#!/bin/bash
# round 2, GPU call 2: the batched runner's parity tests, the whole GPU suite, the launch-set sweep, one bench line
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== batched parity tests"
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_zz_gpu_batched_motionnet.py -x -q 2>&1 | tail -25
echo "== whole GPU suite"
timeout 1200 python -m pytest tests -m gpu -q -x 2>&1 | tail -15
echo "== sweep"
timeout 400 python tools/batch_sweep.py 16 6 2>&1 | grep -v amdgpu.ids
INSMOS_SPLIT_TAP_MOD=0 timeout 300 python tools/batch_sweep.py 16 6 2>&1 | grep -v amdgpu.ids
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_a.json 2> $R/gpurun_out/r02/bench_a.err; tail -c 3000 $R/gpurun_out/r02/bench_a.json; tail -3 $R/gpurun_out/r02/bench_a.err
