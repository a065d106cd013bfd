#!/bin/bash
# per-layer PMC passes of the sparse-conv launches (one rocprofv3 run per counter group; --kernel-trace only)
set -u
R=$(pwd)
export TMPDIR=/tmp
mkdir -p $R/gpurun_out/pmc
python -c "import __graft_entry__ as g; g.build()" >/dev/null 2>&1
i=0
for grp in \
 "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
 "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU" \
 "TA_BUSY_avr TA_BUFFER_LOAD_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" \
 "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum TCC_EA0_RDREQ_sum" ; do
  i=$((i+1))
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $grp -d $R/gpurun_out/pmc/p$i -o p$i --output-format csv -- python $R/tools/pmc_layers.py ) > $R/gpurun_out/pmc/p$i.log 2>&1
  echo "pass $i rc=$?"; tail -2 $R/gpurun_out/pmc/p$i.log
done
find $R/gpurun_out/pmc -name "*.csv" | head -20
python $R/tools/pmc_join.py $R/gpurun_out/pmc > $R/gpurun_out/pmc/joined.csv; head -5 $R/gpurun_out/pmc/joined.csv
