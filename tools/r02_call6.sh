#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_coords.py tests/test_gpu_batched.py "tests/test_gpu_fullsize.py::test_s0_full_size_against_the_oracle" tests/test_train_unet.py -q 2>&1 | tail -25 | cut -c1-300
echo "== whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -8 | cut -c1-300
echo "== layers B=8"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_wlds.csv 2>&1 | grep -v amdgpu.ids | tail -4
INSMOS_WLDS_MIN_TILES=0 timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_nowlds.csv 2>&1 | grep -v amdgpu.ids | tail -2
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_d.json 2> $R/gpurun_out/r02/bench_d.err; tail -c 2500 $R/gpurun_out/r02/bench_d.json; tail -3 $R/gpurun_out/r02/bench_d.err
