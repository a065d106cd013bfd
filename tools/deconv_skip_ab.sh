#!/bin/bash
# insmos_deconv_head_skip (the stack's constant sites skipped in the fused deblock + heads, INSMOS_DECONV_SKIP=1: default) against
# the full launch (=0): bits (BEV tests, model tests), the per-layer table of a launch set of 8, the bench's timed steps, interleaved.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/deconv_skip_ab.txt; : > $O
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 300 python -m pytest -x -q tests/test_gpu_conv.py -k "bev or deconv" 2>&1 | tail -2 | tee -a $O
timeout 600 python -m pytest -x -q tests/test_gpu_model.py tests/test_gpu_batched.py 2>&1 | tail -2 | tee -a $O
for m in 0 1; do
  INSMOS_DECONV_SKIP=$m timeout 200 python tools/batch_layers.py 8 gpurun_out/layers_dskip$m.csv > /dev/null 2>&1
  echo "per-layer (launch set of 8), INSMOS_DECONV_SKIP=$m:" | tee -a $O; grep -E "^(bev5|deconv|TOTAL)" gpurun_out/layers_dskip$m.csv | tee -a $O
done
for m in 0 1 0 1; do
  echo "INSMOS_DECONV_SKIP=$m: $(INSMOS_DECONV_SKIP=$m python bench.py --timed-only --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-60)" | tee -a $O
done
