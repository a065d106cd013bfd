#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE over the convolution launches of the tools/pmc_traffic.sh passes (one launch set of WPL windows)."""
import csv
import glob
import json
import os
import sys

root, tag, wpl = sys.argv[1], sys.argv[2], int(sys.argv[3])
cfg = sys.argv[4] if len(sys.argv) > 4 else "cfg2"
CONV = ("k_sparse_conv", "k_conv_rowlane", "k_conv_row32", "k_conv_tapc", "k_deconv_head", "k_resolve_taps<2, 1, 1>", "k_parent_cubes", "k_const_conv125", "k_bev_conv", "k_bev_group_lists")
tot, allk = {}, {}
launches = 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    # the LAST launch set of the trace only (the run does one warm-up step first: a first call may repeat part of its launches
    # while the arena grows, and computes the BEV constants): everything from the last first-layer kernel on
    starts = [i for i, r in enumerate(rows) if "k_parent_cubes" in r["Kernel_Name"]]
    rows = rows[starts[-1]:] if starts else rows
    conv = [r for r in rows if any(k in r["Kernel_Name"] for k in CONV)]
    tot[c] = sum(float(r["Counter_Value"]) for r in conv) * 1024.0  # counter unit: KB
    allk[c] = sum(float(r["Counter_Value"]) for r in rows) * 1024.0
    launches = len(conv)
# Round 4: FETCH_SIZE calibrated on launches of the gather kernels themselves over tables with a known byte count
# (tools/pmc_gather_layers.py, profiles/r04_pmc_gather_layers.json): factor 1.00 for 8 B/lane and 16 B/lane gathers alike -- a
# gathered row is a 32-64 B piece, i.e. 64-B fabric requests, tallied at face value.  The x2 of MI355X_MICROARCH.md is for 1 KiB-per-wave
# streaming reads (128-B requests); rounds 1-3 applied it to everything and so doubled the fetch side.  The x2 figure stays in the
# file as an upper bound.
hbm = tot["FETCH_SIZE"] + tot["WRITE_SIZE"]
repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:   # the binary the passes ran on: bench.py prints the record only while this is the loaded library's hash (traffic_stale otherwise)
    lib_hash = open(os.path.join(repo, "insmos_amd", "libinsmos_hip.so.srchash")).read().strip()[:12]
except OSError:
    lib_hash = None
out = {
    "lib_source_hash": lib_hash, "config": cfg,
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --timed-only --steps 1 "
               f"--warmup 0 --windows-per-step {wpl} (INSMOS_WINDOWS_IN_FLIGHT=1 INSMOS_WINDOWS_PER_LAUNCH={wpl})",
    "scope": f"all convolution launches of ONE launch set of {wpl} {'cfg-2 S0' if cfg == 'cfg2' else cfg} windows",
    "windows_per_launch": wpl,
    "fetch_size_raw_bytes": tot["FETCH_SIZE"], "fetch_size_x2_upper_bound_bytes": 2 * tot["FETCH_SIZE"],
    "correction": "FETCH_SIZE at face value: calibrated = 1.00 on the gather kernels (64-B requests; tools/pmc_gather_layers.py); "
                  "the guide's x2 applies to 128-B streaming requests only and is kept as an upper bound "
                  "(hbm_bytes_per_window_x2_upper_bound); WRITE_SIZE calibrated = 1.00; counter unit KB",
    "hbm_bytes_per_window_x2_upper_bound": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / wpl,
    "write_size_bytes": tot["WRITE_SIZE"], "hbm_bytes_per_launch_set": hbm, "hbm_bytes_per_window": hbm / wpl,
    "conv_launches_per_launch_set": launches, "hbm_bytes_per_launch": hbm / max(launches, 1),
    "all_kernels_hbm_bytes_per_window": (allk["FETCH_SIZE"] + allk["WRITE_SIZE"]) / wpl,
}
path = os.path.join(repo, "gpurun_out", f"{tag}_pmc_traffic{'' if cfg == 'cfg2' else '_' + cfg}.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
