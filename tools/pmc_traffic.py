#!/usr/bin/env python3
"""Sum FETCH_SIZE / WRITE_SIZE over the conv launches of the LAST window of the tools/pmc_traffic.sh passes."""
import csv
import glob
import json
import os
import sys

root, tag = sys.argv[1], sys.argv[2]
tot = {}
launches = 0
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(os.path.join(root, c, "**", "*counter_collection.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == c]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    q = [i for i, r in enumerate(rows) if "k_quant_keys" in r["Kernel_Name"]]
    last = rows[q[-1]:]  # the last forward of the run
    conv = [r for r in last if "k_sparse_conv" in r["Kernel_Name"] or "k_resolve_taps<2, 1, 1>" in r["Kernel_Name"]]
    tot[c] = sum(float(r["Counter_Value"]) for r in conv) * 1024.0  # counter unit: KB
    launches = len(conv)
out = {
    "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE (separate passes) -- python bench.py --steps 1 --warmup 1 "
               "--windows-per-step 1 --no-cpu-baseline (INSMOS_WINDOWS_IN_FLIGHT=1)",
    "scope": "all sparse-conv launches (k_sparse_conv* + constant-input first layer) of ONE window, cfg-2 S0",
    "fetch_size_raw_bytes": tot["FETCH_SIZE"], "fetch_size_corrected_bytes": 2 * tot["FETCH_SIZE"],
    "correction": "MI355X_MICROARCH.md section HBM: on gfx950 FETCH_SIZE counts 128-B requests at 64 B for 16 B/lane loads -> x2; "
                  "WRITE_SIZE uncalibrated, taken as is; counter unit KB",
    "write_size_bytes": tot["WRITE_SIZE"], "hbm_bytes_per_window": 2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"],
    "launches_per_window": launches, "hbm_bytes_per_launch": (2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]) / max(launches, 1),
}
path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", f"{tag}_pmc_traffic.json")
json.dump(out, open(path, "w"), indent=1)
print(json.dumps(out, indent=1))
