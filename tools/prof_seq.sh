#!/bin/bash
# rocprofv3 kernel stats of bench.py --timed-only, one launch set at a time, second stream off: per-kernel times of the pipeline.
# usage: bash tools/prof_seq.sh r03 name   -> gpurun_out/r03/kstats_<name>.csv   (needs the calibration cache: run bench.py first)
R=$(pwd); TAG=${1:-r03}; NAME=${2:-seq}; O=$R/gpurun_out/$TAG; mkdir -p $O
export TMPDIR=/tmp
rm -rf $O/prof_$NAME
( cd /tmp && INSMOS_TWO_STREAMS=${TS:-0} INSMOS_WINDOWS_IN_FLIGHT=${FL:-1} timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof_$NAME -o prof --output-format csv -- \
    python $R/bench.py --timed-only --steps 4 --warmup 1 ) > $O/rocprof_$NAME.log 2>&1
ST=$(find $O/prof_$NAME -name "*kernel_stats.csv" | head -1)
cp "$ST" $O/kstats_$NAME.csv
rm -rf $O/prof_$NAME
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/kstats_$NAME.csv")))
W = 120.0
conv = non = 0.0
out = []
for r in rows:
    n = r["Name"]; t = int(r["TotalDurationNs"]) / W / 1e3; c = int(r["Calls"]) / 15.0
    isconv = any(k in n for k in ("k_sparse_conv", "k_bev_conv3x3", "k_deconv_head", "k_resolve_taps<2, 1, 1>", "k_conv_lds"))
    conv += t if isconv else 0; non += 0 if isconv else t
    out.append((t, c, isconv, n.split("(")[0][-60:]))
print("conv us/window %.1f   non-conv %.1f   launches/set %.0f" % (conv, non, sum(c for _, c, _, _ in out)))
for t, c, cv, s in sorted(out, key=lambda x: -x[0])[:45]:
    print("%7.1f us/win %6.1f calls/set %s %s" % (t, c, "C" if cv else " ", s))
PY
