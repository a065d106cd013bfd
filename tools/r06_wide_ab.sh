#!/bin/bash
# round 6: the staged 32-row kernel (csrc/spconv_wide.hip) against the chunk-split tiles, per layer, interleaved in one process
# -> gpurun_out/r06_wide/layers_ab_wide_<tag>.csv; then the bitwise test and two bench lines
R=$(pwd); O=$R/gpurun_out/r06_wide; mkdir -p $O
TAG=${1:-v}
timeout 300 python -m pytest tests/test_gpu_conv.py -x -q -m gpu -k "wide" 2>&1 | tail -3
BATCH_LAYERS_ENV="INSMOS_CONV_WIDE=0;INSMOS_CONV_WIDE=1" BATCH_LAYERS_ROUNDS=${2:-3} timeout 600 python tools/batch_layers.py 8 $O/layers_ab_wide_$TAG.csv 2>&1 | grep -v amdgpu.ids | tail -2
for v in 0 1 0 1; do INSMOS_CONV_WIDE=$v timeout 300 python bench.py --timed-only --steps 20 --warmup 3 2>/dev/null | cut -c1-60; done
