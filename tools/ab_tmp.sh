#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_conv.py tests/test_gpu_model.py tests/test_gpu_batched.py -q 2>&1 | tail -5 | cut -c1-300
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_dh.csv 2>&1 | grep -v amdgpu.ids | grep "deconv\|^#"
timeout 300 python tools/batch_layers.py 1 $R/gpurun_out/r02/layers_b1_dh.csv 2>&1 | grep -v amdgpu.ids | grep "deconv\|^#"
