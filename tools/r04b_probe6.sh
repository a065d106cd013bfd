#!/bin/bash
# round 4, second half: list kernel with chunk-major workgroup order; training tests + cfg-5 with the gather backward fixed
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 300 python -m pytest -x -q tests/test_gpu_conv.py -k "bev" 2>&1 | tail -2
timeout 900 python -m pytest -x -q tests/test_train_slice.py tests/test_train_unet.py tests/test_zz_gpu_reference_golden.py 2>&1 | tail -3
timeout 200 python tools/batch_layers.py 8 gpurun_out/layers_list_cm.csv > /dev/null 2>&1
echo "per-layer (launch set of 8), list (chunk-major):"; grep -E "^(bev|TOTAL)" gpurun_out/layers_list_cm.csv
for m in 0 1 0 1; do
  echo "list=$m: $(INSMOS_BEV_SKIP_LIST=$m python bench.py --timed-only --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-50)"
done
echo "mixed seeds list=0: $(INSMOS_BEV_SKIP_LIST=0 python bench.py --timed-only --mixed-seeds --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-50)"
echo "mixed seeds list=1: $(INSMOS_BEV_SKIP_LIST=1 python bench.py --timed-only --mixed-seeds --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-50)"
for g in 0 1; do
INSMOS_GATHER_TORCH=$g timeout 200 python bench.py --config cfg5 --steps 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5_probe.json
python -c "import json; j=json.load(open('gpurun_out/cfg5_probe.json')); print('cfg5 (torch gather=$g) windows/s', j['value'], 'ms/step', j['ms_per_step'], 'loss', j['loss'])"
done
