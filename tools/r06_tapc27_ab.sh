#!/bin/bash
# round 6: the 27-tap tap-split layers of the 3D UNet's level 2 in tap-compacted form (INSMOS_CONV_TAPC bit 3) against the tile kernels
R=$(pwd); O=$R/gpurun_out/r06_tapc27; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "tap_compacted" 2>&1 | tail -3
INSMOS_CONV_TAPC=9 timeout 600 python -m pytest tests/test_gpu_batched.py tests/test_gpu_model.py -x -q -m gpu 2>&1 | tail -3
BATCH_LAYERS_ENV="INSMOS_CONV_TAPC=1;INSMOS_CONV_TAPC=9" BATCH_LAYERS_ROUNDS=3 timeout 600 python tools/batch_layers.py 8 $O/layers_ab_tapc27.csv 2>&1 | grep -v amdgpu.ids | tail -1
awk -F, '$6!=$7 && ($7/$6>1.03 || $7/$6<0.97)' $O/layers_ab_tapc27.csv
for v in 1 9 1 9; do INSMOS_CONV_TAPC=$v timeout 300 python bench.py --timed-only --steps 20 --warmup 3 2>/dev/null | cut -c1-60; done
