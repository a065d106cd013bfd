#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== targeted tests"
timeout 900 python -m pytest tests/test_gpu_batched.py tests/test_data_stage.py tests/test_gpu_coords.py -m gpu -q 2>&1 | tail -8 | cut -c1-300
echo "== sweep W=24"
timeout 400 python tools/batch_sweep.py 24 4 "8x1,8x2,8x3,12x2,6x4,4x3" 2>&1 | grep -v amdgpu.ids | tail -8
echo "== driver"
INSMOS_BENCH_DIR=/dev/shm timeout 300 python tools/driver_bench.py 320 2>&1 | grep "^\[" | head -12
