#!/bin/bash
# round 6: the tap-compacted kernel's variants against the tile kernels, per layer, INTERLEAVED IN ONE PROCESS (tools/batch_layers.py
# BATCH_LAYERS_ENV) -> gpurun_out/r06_tapc/layers_ab_<tag>.csv
R=$(pwd); O=$R/gpurun_out/r06_tapc; mkdir -p $O
V=${1:-"INSMOS_CONV_TAPC=0;INSMOS_CONV_TAPC=1;INSMOS_CONV_TAPC=1,INSMOS_TAPC_ROW32=0;INSMOS_CONV_TAPC=1,INSMOS_TAPC_ROW32=2;INSMOS_CONV_TAPC=3;INSMOS_CONV_TAPC=7"}
TAG=${2:-v}
BATCH_LAYERS_ENV="$V" BATCH_LAYERS_ONLY_K=81 BATCH_LAYERS_ROUNDS=${3:-3} timeout 600 python tools/batch_layers.py 8 $O/layers_ab_$TAG.csv 2>&1 | grep -v amdgpu.ids
