R=$(pwd); O=$R/gpurun_out/r06_bn_camp; mkdir -p $O; export TMPDIR=/tmp
run() { # rows c chunk
  rm -rf $O/prof
  ( cd /tmp && INSMOS_BN_CHUNK=$3 timeout 200 rocprofv3 --kernel-trace --stats -d $O/prof -o p --output-format csv -- python $R/tools/bn_kernel_probe.py $1 $2 ) 2>&1 | grep "^rows" | tr '\n' ' '
  ST=$(find $O/prof -name "*kernel_stats.csv" | head -1)
  python - "$ST" <<'PY'
import csv, sys, re
o=[]
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"k_bnseg_(\w+)_v4", r["Name"])
    if m: o.append("%s %.1f" % (m.group(1), float(r["AverageNs"]) / 1e3))
print(" | ".join(sorted(o)))
PY
}
for ch in 512 496 488 520 1000; do echo "chunk $ch:"; run 300000 256 $ch; done
for ch in 512 496 1000 1024; do echo "chunk $ch:"; run 108000 128 $ch; done
for ch in 4096 3968 4000 2000 2048; do echo "chunk $ch:"; run 1570000 8 $ch; done
for ch in 2048 1984 2000; do echo "chunk $ch:"; run 261000 32 $ch; done
