#!/bin/bash
# PMC counters of ONE kernel family of the timed workload (separate rocprofv3 passes, --kernel-trace only):
#   bash tools/pmc_kernel.sh k_resolve_taps "SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS" ...
R=$(pwd); KERN=$1; shift; export TMPDIR=/tmp
O=$R/gpurun_out/pmc_kernel; mkdir -p $O
i=0
for set in "$@"; do
  i=$((i+1)); rm -rf $O/p$i
  ( cd /tmp && INSMOS_TWO_STREAMS=0 INSMOS_WINDOWS_IN_FLIGHT=1 INSMOS_WINDOWS_PER_LAUNCH=8 timeout 240 rocprofv3 --kernel-trace --pmc $set \
      -d $O/p$i -o p --output-format csv -- python $R/bench.py --timed-only --steps 1 --warmup 1 --windows-per-step 8 ) > $O/p$i.log 2>&1
  python - "$O/p$i" "$KERN" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counters collected:", open(sys.argv[1] + ".log").read()[-600:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    if sys.argv[2] in r["Kernel_Name"]:
        key = r["Kernel_Name"][:70]
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"]); n[(key, r["Counter_Name"])] += 1
for key, cs in acc.items():
    print(key)
    for c, v in cs.items():
        print("   %-28s %16.0f  (%d dispatches)" % (c, v, n[(key, c)]))
PY
done
