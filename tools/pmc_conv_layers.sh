#!/bin/bash
# SQ / TA counters of single conv layers at launch-set size: where do the waves' cycles go?  One rocprofv3 pass per counter
# group (--kernel-trace only).  Averages are per kernel VARIANT over every launch of the process, i.e. they include the one
# single-window forward that builds the tables -- qualitative (wait_any = parked on s_waitcnt / barrier, wait_inst = issue
# stall: MFMA pipe / RAW or a full vector-memory queue, active = issuing).  usage: bash tools/pmc_conv_layers.sh "conv4.1.0,block1.0.conv1,..." [B=8] [tag]
R=$(pwd); L=${1:-conv4.1.0,block1.0.conv1,block7.0.conv2,conv2.1.0,conv3.1.0,bev1}; B=${2:-8}; TAG=${3:-r02}
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_conv; mkdir -p $O
python $R/tools/conv_once.py $L $B 1 > $O/warm.log 2>&1   # caches the window and the calibration
i=0
for G in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" \
         "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
         "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
         "TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $O/p$i
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $G -d $O/p$i -o p --output-format csv -- python $R/tools/conv_once.py $L $B 2 ) > $O/p$i.log 2>&1
  echo "pass $i rc=$? : $G"
done
python - <<PY
import csv, glob, json, collections
out = collections.OrderedDict()
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_sparse_conv" not in k and "k_bev_conv" not in k:
            continue
        k = k.replace("void insmos::", "").split("(")[0]
        d = out.setdefault(k, collections.OrderedDict())
        d.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
res = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in out.items()}
json.dump(res, open("$R/gpurun_out/${TAG}_pmc_conv_layers.json", "w"), indent=1)
for k, d in res.items():
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    print(k)
    print("   wave-cycles split: wait_any %.2f  wait_inst %.2f  active %.2f | inst mix (of wave cycles): vmem %.3f lds %.3f valu %.3f salu %.3f " % (
        d.get("SQ_WAIT_ANY", 0) / wc, d.get("SQ_WAIT_INST_ANY", 0) / wc, d.get("SQ_ACTIVE_INST_ANY", 0) / wc,
        d.get("SQ_ACTIVE_INST_VMEM", 0) / wc, d.get("SQ_ACTIVE_INST_LDS", 0) / wc, d.get("SQ_ACTIVE_INST_VALU", 0) / wc,
        d.get("SQ_ACTIVE_INST_SCA", 0) / wc))
    print("   " + "  ".join("%s=%.3g" % (c, v) for c, v in d.items() if c.startswith(("TA_", "TCP_", "TCC_", "GRBM", "SQ_INSTS", "SQ_WAVES"))))
PY
