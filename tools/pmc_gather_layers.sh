#!/bin/bash
# Per-layer HBM traffic of the small-channel gather layers, calibrated (tools/pmc_gather_layers.py).  One rocprofv3 pass per counter,
# --kernel-trace only (never combined with other trace domains).   usage (fresh GPU box):  bash tools/pmc_gather_layers.sh r04
R=$(pwd); TAG=${1:-r04}
export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gather; mkdir -p $O
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && INSMOS_WINDOWS_IN_FLIGHT=1 timeout 400 rocprofv3 --kernel-trace --pmc $c -d $O/$c -o p --output-format csv -- \
      python $R/tools/pmc_gather_layers.py ) > $O/$c.log 2>&1
  echo "$c rc=$?"; tail -1 $O/$c.log
done
python $R/tools/pmc_gather_layers.py --join $O $TAG > $O/join.log 2>&1; tail -40 $O/join.log
find $O -name "*kernel_trace.csv" -size +20M -delete
