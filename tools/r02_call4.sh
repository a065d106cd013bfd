#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_batched.py tests/test_gpu_model.py "tests/test_gpu_fullsize.py::test_s0_full_size_against_the_oracle" -q 2>&1 | tail -25 | cut -c1-300
echo "== whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | cut -c1-300
echo "== layers B=8 (BEV kernel on)"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_bevk.csv 2>&1 | grep -v amdgpu.ids | grep "bev\|deconv\|TOTAL\|^#"
timeout 300 python tools/batch_layers.py 1 $R/gpurun_out/r02/layers_b1_bevk.csv 2>&1 | grep -v amdgpu.ids | grep "bev\|deconv\|TOTAL\|^#"
echo "== conv tune at launch-set size"
timeout 600 python tools/conv_tune_batched.py 8 2>&1 | grep -v amdgpu.ids
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_c.json 2> $R/gpurun_out/r02/bench_c.err; tail -c 3600 $R/gpurun_out/r02/bench_c.json; tail -3 $R/gpurun_out/r02/bench_c.err
