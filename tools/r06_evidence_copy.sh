#!/bin/bash
# round 6: copy what tools/round_end.sh / r06_cfg5_kstats.sh / pmc_*.sh left under gpurun_out/ into profiles/ (tracked) under r06 names
O=gpurun_out/r06
cp $O/bench.json profiles/r06_bench.json; cp $O/bench_cfg4.json profiles/r06_bench_cfg4.json; cp $O/bench_cfg5.json profiles/r06_bench_cfg5.json
cp $O/bench_profile.json profiles/r06_bench_profile_round.json; cp $O/pytest_gpu.log profiles/r06_pytest_gpu.log; cp $O/smoke.log profiles/r06_smoke.log
cp $O/layers_b8.csv profiles/r06_layers_b8.csv; cp $O/latency_probe.txt profiles/r06_latency_probe.txt; cp $O/driver_bench.txt profiles/r06_driver_bench.txt
cp $O/rocprof_kernel_stats_fl1.csv profiles/r06_rocprof_kernel_stats_sequential.csv; cp $O/rocprof_kernel_stats_fl4.csv profiles/r06_rocprof_kernel_stats_inflight4.csv
cp $O/roofline_fl1.json profiles/r06_roofline_from_rocprof.json; cp $O/roofline_fl4.json profiles/r06_roofline_from_rocprof_inflight4.json
cp gpurun_out/r06_pmc_traffic.json gpurun_out/r06_pmc_traffic_cfg4.json gpurun_out/r06_pmc_mfma.json profiles/
cp gpurun_out/r06_cfg5/cfg5_kernel_stats.csv profiles/r06_rocprof_kernel_stats_cfg5.csv
[ -f $O/layers_b1.csv ] && cp $O/layers_b1.csv profiles/r06_layers_b1.csv
python tools/kstats_rank.py profiles/r06_rocprof_kernel_stats_sequential.csv 20 > profiles/r06_kernel_families.txt
grep -o '"lib_source_hash": "[a-z0-9]*"' profiles/r06_pmc_traffic.json profiles/r06_pmc_traffic_cfg4.json; cut -c1-12 insmos_amd/libinsmos_hip.so.srchash
