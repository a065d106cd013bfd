#!/bin/bash
R=$(pwd); mkdir -p $R/gpurun_out/r02
echo "== whole GPU suite"
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -6 | cut -c1-300
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke 2>&1 | tail -2
echo "== profile round"
bash tools/profile_round.sh r02 8
echo "== layers"
timeout 300 python tools/batch_layers.py 8 $R/gpurun_out/r02/layers_b8_final.csv 2>&1 | grep -v amdgpu.ids | tail -2
