#!/bin/bash
# round 4, second half: list kernel occupancy / TH variants, cfg-5 with the gather backward + cached zero vectors
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -1
timeout 300 python -m pytest -x -q tests/test_gpu_conv.py -k "bev" 2>&1 | tail -2
INSMOS_BEV_LIST_TH=6 timeout 300 python -m pytest -x -q tests/test_gpu_conv.py -k "bev" 2>&1 | tail -2
timeout 900 python -m pytest -x -q tests/test_train_slice.py tests/test_train_unet.py tests/test_zz_gpu_reference_golden.py 2>&1 | tail -3
for cfg in "1 4" "1 6"; do
  set -- $cfg
  INSMOS_BEV_SKIP_LIST=$1 INSMOS_BEV_LIST_TH=$2 timeout 200 python tools/batch_layers.py 8 gpurun_out/layers_list$1_th$2.csv > /dev/null 2>&1
  echo "per-layer (launch set of 8), list=$1 th=$2:"; grep -E "^(bev|TOTAL)" gpurun_out/layers_list$1_th$2.csv
done
for cfg in "0 4" "1 4" "1 6" "0 4" "1 4" "1 6"; do
  set -- $cfg
  echo "list=$1 th=$2: $(INSMOS_BEV_SKIP_LIST=$1 INSMOS_BEV_LIST_TH=$2 python bench.py --timed-only --steps 10 --warmup 3 2>/dev/null | tail -1 | cut -c1-50)"
done
timeout 200 python bench.py --config cfg5 --steps 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5_probe.json
python -c "import json; j=json.load(open('gpurun_out/cfg5_probe.json')); print('cfg5 windows/s', j['value'], 'ms/step', j['ms_per_step'], 'loss', j['loss'])"
