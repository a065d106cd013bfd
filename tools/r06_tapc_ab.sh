#!/bin/bash
# round 6: the tap-compacted 81-tap kernel (csrc/spconv_tapc.hip) -- parity, then per-layer A/B and the bench line, one gpurun call
#   bash tools/r06_tapc_ab.sh [modes="0 1 3 7"]   -> gpurun_out/r06_tapc/
R=$(pwd); O=$R/gpurun_out/r06_tapc; mkdir -p $O
MODES=${1:-"0 1 3 7"}
echo "== tests"
timeout 900 python -m pytest tests/test_gpu_conv.py -m gpu -q -x --timeout 300 -k "tap_compacted" > $O/pytest_tapc.log 2>&1; tail -3 $O/pytest_tapc.log | cut -c1-300
timeout 900 python -m pytest tests/test_gpu_model.py tests/test_gpu_batched.py -m gpu -q -x --timeout 400 > $O/pytest_model.log 2>&1; tail -3 $O/pytest_model.log | cut -c1-300
for m in $MODES; do
  echo "== layers INSMOS_CONV_TAPC=$m"
  INSMOS_CONV_TAPC=$m timeout 300 python tools/batch_layers.py 8 $O/layers_b8_tapc$m.csv 2>&1 | grep -v amdgpu.ids | tail -1
done
for m in $MODES; do
  echo "== bench --timed-only INSMOS_CONV_TAPC=$m"
  INSMOS_CONV_TAPC=$m timeout 300 python bench.py --timed-only --steps 20 --warmup 3 2> $O/bench_tapc$m.err | tail -1 > $O/bench_tapc$m.json; cut -c1-160 $O/bench_tapc$m.json
done
