#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests"
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_batched.py tests/test_gpu_model.py tests/test_gpu_fullsize.py -q 2>&1 | tail -25 | cut -c1-300
echo "== layers B=8"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_f.csv 2>&1 | grep -v amdgpu.ids | grep "bev\|TOTAL\|^#"
timeout 300 python tools/batch_layers.py 1 $R/gpurun_out/r02/layers_b1_f.csv 2>&1 | grep -v amdgpu.ids | grep "bev\|TOTAL\|^#"
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_f.json 2> $R/gpurun_out/r02/bench_f.err; tail -c 2500 $R/gpurun_out/r02/bench_f.json; tail -3 $R/gpurun_out/r02/bench_f.err
