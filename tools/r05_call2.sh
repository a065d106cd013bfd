#!/bin/bash
O=gpurun_out/r05b; mkdir -p $O
echo "== bench --no-extras (faulthandler)"
timeout 600 python -X faulthandler bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench1.out 2> $O/bench1.err; echo "rc=$?"
tail -c 600 $O/bench1.out; echo; grep -v amdgpu.ids $O/bench1.err | tail -40
