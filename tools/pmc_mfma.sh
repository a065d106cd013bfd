#!/bin/bash
# MFMA-pipe utilisation of the conv launches from PMC counters (one rocprofv3 pass, --kernel-trace only; one window in flight)
R=$(pwd); TAG=${1:-r03}; WPL=${2:-4}
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc_mfma
( cd /tmp && INSMOS_TWO_STREAMS=0 INSMOS_WINDOWS_IN_FLIGHT=1 INSMOS_WINDOWS_PER_LAUNCH=$WPL timeout 240 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES \
    -d $R/gpurun_out/pmc_mfma/p -o p --output-format csv -- python $R/bench.py --timed-only --steps 1 --warmup 1 --windows-per-step $WPL ) > $R/gpurun_out/pmc_mfma/p.log 2>&1
echo "rc=$?"
python - <<PY
import csv, glob, json, os
f = glob.glob("$R/gpurun_out/pmc_mfma/p/**/*counter_collection.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Dispatch_Id"]))
starts = [i for i, r in enumerate(rows) if "k_parent_cubes" in r["Kernel_Name"]]   # the LAST launch set of the trace only
rows = rows[starts[-1]:] if starts else rows
conv = [r for r in rows if any(k in r["Kernel_Name"] for k in ("k_sparse_conv", "k_conv_rowlane", "k_conv_row32", "k_conv_tapc", "k_conv_wide", "k_deconv_head", "k_bev_conv", "k_resolve_taps<2, 1, 1>", "k_parent_cubes", "k_const_conv125"))]
tot = {}
for r in conv:
    tot[r["Counter_Name"]] = tot.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
disp = len({r["Dispatch_Id"] for r in conv})
gui = tot.get("GRBM_GUI_ACTIVE", 0.0)
out = {"command": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA SQ_WAVES -- bench.py --timed-only (one launch set)",
       "scope": "conv launches (k_sparse_conv*, k_bev_conv3x3, k_deconv_head, constant-input first layer) of ONE launch set of $WPL cfg-2 S0 windows (bench.py --timed-only; the profiler serialises kernels)", "windows_per_launch": $WPL,
       "launches": disp, "totals": tot,
       "mfma_busy_over_gpu_active": tot.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024 / 8) if gui else None,
       "note": "GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs; SQ_VALU_MFMA_BUSY_CYCLES = 32 cycles per v_mfma_f32_16x16x4_f32 "
               "(summed over SIMDs). Includes MFMA work on absent rows / padded channels; per-launch profiling overhead (~12 us) inflates "
               "GUI_ACTIVE of the short launches."}
json.dump(out, open("$R/gpurun_out/${TAG}_pmc_mfma.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1500])
PY
