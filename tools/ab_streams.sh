#!/bin/bash
# A/B of the second stream inside a launch set (INSMOS_TWO_STREAMS) against launch sets in flight, on the GPU box.
# usage: bash tools/ab_streams.sh r03   -> gpurun_out/r03/ab_*.json
R=$(pwd); TAG=${1:-r03}; O=$R/gpurun_out/$TAG; mkdir -p $O
python bench.py --steps 10 --warmup 3 2> $O/ab_full_ts1.err | tail -1 > $O/ab_full_ts1.json
INSMOS_TWO_STREAMS=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2> $O/ab_full_ts0.err | tail -1 > $O/ab_full_ts0.json
for TS in 1 0; do for FL in 1 2 3 4; do
  INSMOS_TWO_STREAMS=$TS INSMOS_WINDOWS_IN_FLIGHT=$FL python bench.py --timed-only --steps 10 --warmup 3 --windows-per-step $((8 * (FL > 3 ? 4 : 3))) 2>/dev/null | tail -1 > $O/ab_ts${TS}_fl${FL}.json
  echo "two_streams=$TS in_flight=$FL $(cat $O/ab_ts${TS}_fl${FL}.json)"
done; done
python - <<PY
import json
for t in (1, 0):
    d = json.load(open("$O/ab_full_ts%d.json" % t))
    print("two_streams", t, {k: d.get(k) for k in ("value", "single_window_latency_ms", "value_b1", "value_s0_only")}, d["roofline"]["frac"], d["device_ms_per_window_sum"])
PY
