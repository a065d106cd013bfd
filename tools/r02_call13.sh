#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_batched.py -m gpu -q 2>&1 | tail -6 | cut -c1-300
echo "== layers B=8, XCD remap on / off"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_xcd1.csv 2>&1 | grep -v amdgpu.ids | tail -2
INSMOS_XCD_REMAP=0 timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_xcd0.csv 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $R/gpurun_out/r02/bench_h.json 2> $R/gpurun_out/r02/bench_h.err; head -c 200 $R/gpurun_out/r02/bench_h.json; tail -3 $R/gpurun_out/r02/bench_h.err
