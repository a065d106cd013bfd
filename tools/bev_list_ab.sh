#!/bin/bash
# The skipping BEV layers over compacted row-group lists (k_bev_conv3x3_list, INSMOS_BEV_SKIP_LIST=1: the default for launch sets of
# four windows or more) against the fixed 16 x 4 patches (k_bev_conv3x3<SKIP>, =0): bits (the BEV tests + the model tests in list
# mode), the per-layer table of a launch set of 8, and the bench's timed steps, interleaved.  -> gpurun_out/bev_list_ab.txt
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
O=gpurun_out/bev_list_ab.txt; : > $O
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 300 python -m pytest -x -q tests/test_gpu_conv.py -k "bev" 2>&1 | tail -1 | tee -a $O
INSMOS_BEV_SKIP_LIST=1 timeout 600 python -m pytest -x -q tests/test_gpu_model.py tests/test_gpu_batched.py 2>&1 | tail -1 | tee -a $O
for m in 0 1; do
  INSMOS_BEV_SKIP_LIST=$m timeout 200 python tools/batch_layers.py 8 gpurun_out/layers_list$m.csv > /dev/null 2>&1
  echo "per-layer (launch set of 8), list=$m:" | tee -a $O; grep -E "^(bev|deconv|TOTAL)" gpurun_out/layers_list$m.csv | tee -a $O
done
for m in 0 1 0 1; do
  echo "INSMOS_BEV_SKIP_LIST=$m: $(INSMOS_BEV_SKIP_LIST=$m python bench.py --timed-only --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-60)" | tee -a $O
done
