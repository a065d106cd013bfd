#!/bin/bash
# round 6: probe builds of the tap-compacted kernel (INSMOS_TAPC_DBG: 1 = accumulators in registers, 2 = no gathers, 4 = no MFMAs,
# 8 = no weight loads; sums combine) per layer at launch-set size -> gpurun_out/r06_tapc_probe/
R=$(pwd); O=$R/gpurun_out/r06_tapc_probe; mkdir -p $O
DBGS=${1:-"0 1 2 4 8 10 11"}
for d in $DBGS; do
  INSMOS_CONV_TAPC=1 INSMOS_TAPC_DBG=$d timeout 300 python tools/batch_layers.py 8 $O/layers_dbg$d.csv > $O/log_dbg$d.txt 2>&1
  echo "== dbg $d: $(tail -1 $O/log_dbg$d.txt)"
  grep -E "^block(3|6)|^block7.0.conv1" $O/layers_dbg$d.csv | cut -d, -f1,5,6 | tr '\n' ' '; echo
done
