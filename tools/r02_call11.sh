#!/bin/bash
echo "== sweep W=48"
timeout 500 python tools/batch_sweep.py 48 3 "8x3,8x4,8x6,12x4,12x2,6x6" 2>&1 | grep -v amdgpu.ids | tail -8
echo "== sweep W=24"
timeout 400 python tools/batch_sweep.py 24 4 "8x3,12x2,6x4" 2>&1 | grep -v amdgpu.ids | tail -4
