#!/bin/bash
# knob A/B at launch-set size, one process per setting (the knobs are read once per process): per-layer tables of ONE set of 8 windows
O=gpurun_out/r05d; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python tools/batch_layers.py 8 $O/layers_$tag.csv 2>&1 | grep "^# B\|Error\|error" | sed "s/^/$tag: /"; }
run base A=1
run half4096 INSMOS_CONV_SPLIT_HALF=4096
run half8192 INSMOS_CONV_SPLIT_HALF=8192
run work50 INSMOS_CONV_SPLIT_WORK=50
run ring4 INSMOS_CONV_RING=4
run ring5 INSMOS_CONV_RING=5
run tapmod0 INSMOS_SPLIT_TAP_MOD=0
run base2 A=1
python - <<'PY'
import csv,glob,os
O="gpurun_out/r05d"
tabs={}
for f in sorted(glob.glob(O+"/layers_*.csv")):
    t=os.path.basename(f)[7:-4]
    tabs[t]={r["layer"]:float(r["us"]) for r in csv.DictReader(open(f))}
names=list(tabs["base"].keys())
order=["base","base2","half4096","half8192","work50","ring4","ring5","tapmod0"]
print("%-30s"%"layer"+"".join("%10s"%t for t in order))
for n in names:
    b=tabs["base"][n]
    row=[tabs[t].get(n,0) for t in order]
    if max(abs(x-b) for x in row)>0.03*b+2 or n=="TOTAL":
        print("%-30s"%n+"".join("%10.1f"%x for x in row))
PY
