#!/bin/bash
# round 4, second half: the compacted row-group BEV kernel (INSMOS_BEV_SKIP_LIST=1) against the patch kernel: bits, per-layer times, bench
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 300 python -m pytest -x -q tests/test_gpu_conv.py -k "bev" 2>&1 | tail -3
INSMOS_BEV_SKIP_LIST=1 timeout 600 python -m pytest -x -q tests/test_gpu_model.py tests/test_gpu_batched.py 2>&1 | tail -3
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --sustain-seconds 0 2>/dev/null | tail -1 | cut -c1-160
for m in 1 0 1 0; do
  echo "INSMOS_BEV_SKIP_LIST=$m: $(INSMOS_BEV_SKIP_LIST=$m python bench.py --timed-only --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-120)"
done
for m in 0 1; do
  INSMOS_BEV_SKIP_LIST=$m timeout 200 python tools/batch_layers.py 8 gpurun_out/layers_list$m.csv > /dev/null 2>&1
  echo "per-layer (launch set of 8), list=$m:"; grep -E "^(bev|deconv|TOTAL)" gpurun_out/layers_list$m.csv
done
INSMOS_BEV_SKIP_LIST=1 timeout 120 python tools/b1_ab.py 2>&1 | grep "chain 1, bev cosplit 1024" 
