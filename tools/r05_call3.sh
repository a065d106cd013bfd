#!/bin/bash
O=gpurun_out/r05c; mkdir -p $O
echo "== whole GPU suite"
timeout 1500 python -m pytest tests -m gpu -q --timeout 500 > $O/pytest_gpu.log 2>&1; tail -4 $O/pytest_gpu.log | cut -c1-300
echo "== bench default (tail -1 must be the JSON line)"
timeout 1200 python bench.py --steps 20 --warmup 3 > $O/bench.out 2> $O/bench.err; echo "rc=$?"
tail -1 $O/bench.out > $O/bench.json; head -c 200 $O/bench.json; echo
python - <<'PY'
import json
j=json.load(open("gpurun_out/r05c/bench.json"))
print({k:j.get(k) for k in ("value","value_b1","single_window_latency_ms","value_mixed_seeds","rccl_world1_ok","launches_per_window","device_ms_per_window_sum")})
print("extras", json.dumps(j.get("extras"))[:1500]); print("frac", j["roofline"]["frac"], j["roofline"]["kernel_ms_per_window"])
PY
echo "== multi-table A/B (B=1 latency + set rate)"
for m in 0 1; do INSMOS_TABLES3D_MULTI=$m timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline --sustain-seconds 0 2>/dev/null | tail -1 | python -c "
import json,sys; j=json.loads(sys.stdin.read()); print('multi=$m', j['value'], j['value_b1'], j['single_window_latency_ms'], j['launches_per_window'], j['kernel_ms_per_window'].get('build_nbr'))"; done
