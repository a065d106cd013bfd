#!/bin/bash
# BatchNorm chunk length (INSMOS_BN_CHUNK) against the training step: one bench.py --config cfg5 run per setting
for c in "$@"; do
  INSMOS_BN_CHUNK=$c timeout 200 python bench.py --config cfg5 --steps 6 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/bn_probe.json
  python -c "import json; j=json.load(open('/tmp/bn_probe.json')); print('chunk', '$c', 'windows/s', j['value'], 'ms/step', j['ms_per_step'], 'batchnorm ms', j['kernel_ms_per_step']['batchnorm'])"
done
