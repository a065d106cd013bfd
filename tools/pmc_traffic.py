#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE over the convolution launches of the tools/pmc_traffic.sh passes (one launch set of WPL windows)."""
import csv
import glob
import json
import os
import sys

root, tag, wpl = sys.argv[1], sys.argv[2], int(sys.argv[3])
CONV = ("k_sparse_conv", "k_deconv_head", "k_resolve_taps<2, 1, 1>", "k_parent_cubes", "k_const_conv125", "k_bev_conv")
tot, allk = {}, {}
launches = 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    conv = [r for r in rows if any(k in r["Kernel_Name"] for k in CONV)]
    tot[c] = sum(float(r["Counter_Value"]) for r in conv) * 1024.0  # counter unit: KB
    allk[c] = sum(float(r["Counter_Value"]) for r in rows) * 1024.0
    launches = len(conv)
hbm = 2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --timed-only --steps 1 "
               f"--warmup 0 --windows-per-step {wpl} (INSMOS_WINDOWS_IN_FLIGHT=1 INSMOS_WINDOWS_PER_LAUNCH={wpl})",
    "scope": f"all convolution launches of ONE launch set of {wpl} cfg-2 S0 windows",
    "windows_per_launch": wpl,
    "fetch_size_raw_bytes": tot["FETCH_SIZE"], "fetch_size_corrected_bytes": 2 * tot["FETCH_SIZE"],
    "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane loads -> x2; "
                  "WRITE_SIZE uncalibrated, taken as is; counter unit KB",
    "write_size_bytes": tot["WRITE_SIZE"], "hbm_bytes_per_launch_set": hbm, "hbm_bytes_per_window": hbm / wpl,
    "conv_launches_per_launch_set": launches, "hbm_bytes_per_launch": hbm / max(launches, 1),
    "all_kernels_hbm_bytes_per_window": (2 * allk["FETCH_SIZE"] + allk["WRITE_SIZE"]) / wpl,
}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"{tag}_pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
