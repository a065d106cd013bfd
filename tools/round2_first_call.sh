#!/bin/bash
# First GPU call of round 2 (everything round 1 left unmeasured when its GPU budget ended), ~3 GPU-minutes:
#   1. the whole training test group with the MFMA dW kernel selected  -> is it correct everywhere it is used?
#   2. the full training step timed with the LDS dW kernel and with the MFMA one (tools/train_step_bench.py)
#   3. the MotionNet step alone, both kernels (round-1 figure: 24.2 ms with the LDS kernel)
#   4. the driver end to end (tools/driver_bench.py) -- round-1 figure 323 scans/s
# usage (repo root on the GPU box):  bash tools/round2_first_call.sh  > gpurun_out/round2_first.log 2>&1
R=$(pwd); mkdir -p $R/gpurun_out
echo "== 1. training tests, INSMOS_DW_MFMA=1"
INSMOS_DW_MFMA=1 timeout 400 python -m pytest tests/test_train_slice.py tests/test_train_unet.py -q -m gpu 2>&1 | tail -4
echo "== 1b. whole GPU suite, nothing staged"
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -30
echo "== 1c. batched MotionNet against one window after the other"
timeout 200 python tools/batched_motionnet_probe.py 4 2>&1 | tail -2
echo "== 2. full training step (S0 window)"
timeout 300 python tools/train_step_bench.py 2>&1 | tail -14
INSMOS_DW_MFMA=1 timeout 300 python tools/train_step_bench.py 2>&1 | tail -14
echo "== 3. MotionNet step"
timeout 200 python tools/train_motionnet_bench.py 2>&1 | tail -1
INSMOS_DW_MFMA=1 timeout 200 python tools/train_motionnet_bench.py 2>&1 | tail -1
echo "== 4. driver"
INSMOS_BENCH_DIR=/dev/shm timeout 300 python tools/driver_bench.py 300 2>&1 | grep "^\[" | head -8
