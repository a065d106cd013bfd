#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== tests"
timeout 1200 python -m pytest tests/test_gpu_conv.py tests/test_gpu_batched.py tests/test_gpu_model.py "tests/test_gpu_fullsize.py::test_s0_full_size_against_the_oracle" -q 2>&1 | tail -8 | cut -c1-300
echo "== layers B=8: 32-row split tiles on / off"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_s2on.csv 2>&1 | grep -v amdgpu.ids | tail -1
INSMOS_SPLIT2_MIN_GROUPS=0 timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_s2off.csv 2>&1 | grep -v amdgpu.ids | tail -1
