#!/bin/bash
# round-5 evidence in one call: whole GPU suite + smoke + the default bench line, then the profile half, then the calibrated
# per-layer PMC table.  Everything lands under gpurun_out/r05/ (+ gpurun_out/r05_pmc_gather_layers.json).
bash tools/round_end.sh r05 all 2>&1 | cut -c1-400
echo "== per-layer HBM counters (calibrated)"
bash tools/pmc_gather_layers.sh r05 2>&1 | tail -45 | cut -c1-300
