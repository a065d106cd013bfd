#!/bin/bash
# round 6: where the staged 32-row kernel's time goes (128 -> 128 layers; PROBE builds drop gathers / MFMAs / weight loads)
R=$(pwd); O=$R/gpurun_out/r06_wide; mkdir -p $O
# the probe instantiations are not in the product library: build one with them first (its own source hash), and rebuild the product after
INSMOS_EXTRA_HIPCC_FLAGS=-DINSMOS_WIDE_PROBE_BUILD python -c "import __graft_entry__ as g; g.build_product()" 2>&1 | tail -1
BATCH_LAYERS_ENV="INSMOS_CONV_WIDE=0;INSMOS_CONV_WIDE=1;INSMOS_CONV_WIDE=1,INSMOS_WIDE_PROBE=1;INSMOS_CONV_WIDE=1,INSMOS_WIDE_PROBE=2;INSMOS_CONV_WIDE=1,INSMOS_WIDE_PROBE=3;INSMOS_CONV_WIDE=1,INSMOS_WIDE_PROBE=4;INSMOS_CONV_WIDE=1,INSMOS_WIDE_PROBE=5;INSMOS_CONV_WIDE=1,INSMOS_WIDE_PROBE=6" BATCH_LAYERS_ROUNDS=3 timeout 600 python tools/batch_layers.py 8 $O/layers_probe_wide.csv 2>&1 | grep -v amdgpu.ids | tail -2
grep -E "^layer|,128,128," $O/layers_probe_wide.csv
python -c "import __graft_entry__ as g; g.build_product()" 2>&1 | tail -1
