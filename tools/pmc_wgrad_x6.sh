#!/bin/bash
# PMC passes over the split-bf16 weight-gradient micro-benchmark (diagnostic).  Usage: bash tools/pmc_wgrad_x6.sh  (on the GPU box)
set -e
R=$PWD
export LD_LIBRARY_PATH=$R/adm_amd
hipcc -O3 -std=c++17 --offload-arch=gfx950 -Wno-unused-value tools/bench_wgrad_x6.cpp -Ladm_amd -ladm_hip -o /tmp/bwx 2>/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_wx6/p$i -- /tmp/bwx > /dev/null 2>&1 || echo "pass $i failed: $grp"
done
cd $R
for j in 1 2 3 4 5 6 7 8; do echo "== pass $j"; python tools/pmc_kernels.py gpurun_out/pmc_wx6/p$j 300 2>/dev/null | head -3; done
