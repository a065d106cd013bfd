#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests (compact C=8/4 weight fragments)"
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_batched.py tests/test_train_slice.py tests/test_train_unet.py -m gpu -q 2>&1 | tail -8 | cut -c1-300
echo "== layers B=8"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_g.csv 2>&1 | grep -v amdgpu.ids | grep "K81,8\|,8,8,\|conv_input\|TOTAL\|^#\|conv1p1s2\|conv2p2s2"
echo "== bench"
timeout 600 python bench.py --steps 10 --warmup 3 > $R/gpurun_out/r02/bench_g.json 2> $R/gpurun_out/r02/bench_g.err; tail -c 2300 $R/gpurun_out/r02/bench_g.json | head -c 700; tail -3 $R/gpurun_out/r02/bench_g.err
