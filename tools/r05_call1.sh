#!/bin/bash
# round 5, first GPU call: the RCCL world-1 tests, the changed tests, smoke, the probe tables for the Cout 16-32 layers, the default bench line
O=gpurun_out/r05a; mkdir -p $O
echo "== rccl + two ranks + changed tests"
timeout 900 python -m pytest tests/test_zz_gpu_rccl_world1.py tests/test_zz_gpu_two_ranks.py -q -x --timeout 600 > $O/pytest_rccl.log 2>&1; tail -5 $O/pytest_rccl.log | cut -c1-400
timeout 600 python -m pytest tests/test_train_unet.py tests/test_train_slice.py -q -m gpu -k "batched_training or segmented_batchnorm" --timeout 400 > $O/pytest_bn.log 2>&1; tail -3 $O/pytest_bn.log | cut -c1-400
echo "== smoke"
timeout 300 python __graft_entry__.py --smoke > $O/smoke.log 2>&1; tail -2 $O/smoke.log | cut -c1-300
echo "== conv probe (set of 8)"
timeout 400 python tools/conv_probe.py block6.0.conv1,block6.0.conv2,block7.0.conv1,block3.0.conv2,conv2.1.0,conv_up_m2.0,conv_up_instance_block_up3.0,inv_conv2.0 8 2>&1 | grep -v amdgpu.ids > $O/conv_probe_b8.txt; cat $O/conv_probe_b8.txt | cut -c1-260
echo "== conv tune (set of 8)"
timeout 400 python tools/conv_tune_batched.py 8 block6.0.conv1,block7.0.conv1,block3.0.conv2,conv2.1.0,conv_up_m1.0 2>&1 | grep -v amdgpu.ids > $O/conv_tune_b8.txt; cat $O/conv_tune_b8.txt | cut -c1-400
echo "== bench (default line, with extras)"
timeout 900 python bench.py --steps 20 --warmup 3 2> $O/bench.err | tail -1 > $O/bench.json; cut -c1-400 $O/bench.json; tail -3 $O/bench.err | cut -c1-300
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05a/bench.json"))
print({k:j.get(k) for k in ("value","value_b1","single_window_latency_ms","value_mixed_seeds","rccl_world1_ok","launches_per_window","device_ms_per_window_sum")})
print("rccl", j.get("rccl_world1")); print("extras", j.get("extras")); print("frac", j["roofline"]["frac"], j["roofline"]["kernel_ms_per_window"])
print(j.get("kernel_ms_per_window"))
PY
