#!/bin/bash
# round 4, second half: first GPU pass over the level-down chain, the BEV channel split and the BatchNorm loads-in-flight change
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build()" 2>&1 | tail -2
timeout 600 python -m pytest -x -q tests/test_gpu_coords.py::test_level_down_chain_equals_the_per_level_calls "tests/test_gpu_conv.py" -k "chain or bev" 2>&1 | tail -5
timeout 600 python -m pytest -x -q tests/test_gpu_model.py tests/test_train_slice.py 2>&1 | tail -5
timeout 300 python tools/b1_ab.py 2>&1 | grep -v Warning | tee gpurun_out/b1_ab.txt | tail -10
timeout 200 python bench.py --config cfg5 --steps 6 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/cfg5_probe.json
python -c "import json; j=json.load(open('gpurun_out/cfg5_probe.json')); print('cfg5 windows/s', j['value'], 'ms/step', j['ms_per_step'], 'kernel ms', j.get('kernel_ms_per_step'))"
timeout 200 python tools/bn_shape_probe.py 2>&1 | grep -v Warning | tee gpurun_out/bn_shape_probe.txt | tail -14
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/b1trace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/b1trace -o b1 -- python $GRAFT_REPO_ROOT/tools/b1_trace.py 2>&1 | grep "windows:" 
cd $GRAFT_REPO_ROOT && python tools/b1_trace.py --analyze /tmp/b1trace > gpurun_out/b1_trace_after.txt 2>&1; head -32 gpurun_out/b1_trace_after.txt
