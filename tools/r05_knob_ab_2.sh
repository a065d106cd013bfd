#!/bin/bash
O=gpurun_out/r05e; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python tools/batch_layers.py 8 $O/layers_$tag.csv 2>&1 | grep "^# B\|Error\|error" | sed "s/^/$tag: /"; }
run base A=1
run quarter4096 INSMOS_CONV_SPLIT_QUARTER=4096
run c64half8192 INSMOS_CONV_SPLIT_HALF_C64=8192
run old1536 INSMOS_CONV_SPLIT_HALF=1536
run bevlist0 INSMOS_BEV_SKIP_LIST=0
run inflight INSMOS_TWO_STREAMS=15
python - <<'PY'
import csv,glob,os
O="gpurun_out/r05e"
tabs={}
for f in sorted(glob.glob(O+"/layers_*.csv")):
    t=os.path.basename(f)[7:-4]
    tabs[t]={r["layer"]:float(r["us"]) for r in csv.DictReader(open(f))}
names=list(tabs["base"].keys())
order=[t for t in ["base","old1536","quarter4096","c64half8192","bevlist0","inflight"] if t in tabs]
print("%-30s"%"layer"+"".join("%12s"%t for t in order))
for n in names:
    b=tabs["base"][n]
    row=[tabs[t].get(n,0) for t in order]
    if max(abs(x-b) for x in row)>0.04*b+2 or n=="TOTAL":
        print("%-30s"%n+"".join("%12.1f"%x for x in row))
PY
for ws in "8 4" "16 2" "12 3"; do set -- $ws; INSMOS_WINDOWS_PER_LAUNCH=$1 INSMOS_WINDOWS_IN_FLIGHT=$2 timeout 200 python bench.py --timed-only --steps 10 --warmup 2 --windows-per-step $(( $1 * $2 )) 2>/dev/null | tail -1 | cut -c1-200; done
