#!/bin/bash
timeout 600 python -m pytest tests/test_gpu_coords.py -q -m gpu -k "quantize_levels" --timeout 300 2>&1 | tail -2
timeout 900 python -m pytest tests/test_train_slice.py -q -m gpu -k "motionnet_training" --timeout 600 2>&1 | tail -2
for v in 0 1; do INSMOS_NBR125_RESOLVER=$v timeout 300 python bench.py --config cfg5 --steps 8 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('resolver=$v', j['ms_per_step'], j['kernel_ms_per_step'].get('build_nbr'))"; done
