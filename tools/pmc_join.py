#!/usr/bin/env python3
"""Join the counter_collection CSVs of tools/pmc_layers.sh: one row per sparse-conv launch of the LAST window."""
import csv
import glob
import json
import os
import sys

root = sys.argv[1]
order = json.load(open(os.path.join(os.path.dirname(root.rstrip("/")), "conv_order.json")))
n = len(order)
cols, table = [], [dict() for _ in range(n)]
for f in sorted(glob.glob(os.path.join(root, "p*", "**", "*counter_collection.csv"), recursive=True)):
    per = {}  # dispatch id -> {counter: value}
    names = {}
    for r in csv.DictReader(open(f)):
        if "k_sparse_conv" not in r["Kernel_Name"]:
            continue
        d = int(r["Dispatch_Id"])
        per.setdefault(d, {})[r["Counter_Name"]] = float(r["Counter_Value"])
        names[d] = (r["Kernel_Name"], r.get("Grid_Size", ""), r.get("VGPR_Count", r.get("Arch_VGPR_Count", "")))
    ds = sorted(per)[-n:]
    for i, d in enumerate(ds):
        table[i].update(per[d])
        table[i]["kernel"] = names[d][0].split("k_sparse_conv")[1][:28]
        table[i]["grid"] = names[d][1]
    for c in per[ds[0]]:
        if c not in cols:
            cols.append(c)
w = csv.writer(sys.stdout)
w.writerow(["layer", "K", "cin", "cout", "n_out", "kernel", "grid"] + cols)
for i, o in enumerate(order):
    w.writerow(list(o) + [table[i].get("kernel", ""), table[i].get("grid", "")] + [table[i].get(c, "") for c in cols])
