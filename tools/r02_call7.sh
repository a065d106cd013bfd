#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_batched.py tests/test_gpu_model.py "tests/test_gpu_fullsize.py::test_s0_full_size_against_the_oracle" -q 2>&1 | tail -25 | cut -c1-300
echo "== layers B=8 (compaction on / off)"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_cmp.csv 2>&1 | grep -v amdgpu.ids | tail -3
INSMOS_CONV_COMPACT=0 timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_nocmp.csv 2>&1 | grep -v amdgpu.ids | tail -2
timeout 300 python tools/batch_layers.py 1 $R/gpurun_out/r02/layers_b1_cmp.csv 2>&1 | grep -v amdgpu.ids | tail -2
INSMOS_CONV_COMPACT=0 timeout 300 python tools/batch_layers.py 1 $R/gpurun_out/r02/layers_b1_nocmp.csv 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_e.json 2> $R/gpurun_out/r02/bench_e.err; tail -c 2500 $R/gpurun_out/r02/bench_e.json; tail -3 $R/gpurun_out/r02/bench_e.err
