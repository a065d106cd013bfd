#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_head.py tests/test_gpu_model.py tests/test_gpu_batched.py tests/test_zz_gpu_reference_golden.py tests/test_refine_stage.py -q -m gpu --timeout 400 2>&1 | tail -3
for c in 0 1; do INSMOS_ONEHOT_COVER=$c timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('cover=$c', j['value'], j['value_b1'], j['single_window_latency_ms'], j['kernel_ms_per_window'].get('boxes_to_onehot'))"; done
