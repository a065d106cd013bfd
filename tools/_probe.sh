#!/bin/bash
O=gpurun_out/r05i; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python tools/batch_layers.py 8 $O/layers_$tag.csv 2>&1 | grep "^# B\|Error\|error" | sed "s/^/$tag: /"; }
run newdef A=1
run ring2 INSMOS_CONV_RING=-1
run full_r3 INSMOS_CONV_SPLIT_HALF=1536 INSMOS_CONV_RING=13
run newdef2 A=1
python - <<'PY'
import csv,glob,os
O="gpurun_out/r05i"
tabs={}
for f in sorted(glob.glob(O+"/layers_*.csv")):
    t=os.path.basename(f)[7:-4]
    tabs[t]={r["layer"]:float(r["us"]) for r in csv.DictReader(open(f))}
order=("newdef","newdef2","ring2","full_r3")
print("%-30s"%"layer"+"".join("%10s"%t for t in order))
for n in tabs["newdef"]:
    b=tabs["newdef"][n]; row=[tabs[t].get(n,0) for t in order]
    if max(abs(x-b) for x in row)>0.03*b+2 or n=="TOTAL": print("%-30s"%n+"".join("%10.1f"%x for x in row))
PY
