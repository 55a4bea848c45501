#!/bin/bash
# PMC passes over the split-bf16 micro-benchmark (diagnostic).  Usage: bash tools/pmc_x6.sh  (on the GPU box)
set -e
R=$PWD
export LD_LIBRARY_PATH=$R/adm_amd
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DW2_QUICK tools/bench_wino2d_x6.cpp -Ladm_amd -ladm_hip -o /tmp/bx 2>/dev/null
hipcc -O3 -std=c++17 --offload-arch=gfx950 -DW2_QUICK tools/bench_wino2d.cpp -Ladm_amd -ladm_hip -o /tmp/bw 2>/dev/null
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  i=$((i+1))
  for b in bx bw; do
    rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc_x6/$b$i -- /tmp/$b > /dev/null 2>&1 || echo "pass $i $b failed: $grp"
  done
done
cd $R
for b in bx bw; do for j in 1 2 3 4 5 6 7 8 9; do echo "== $b pass $j"; python tools/pmc_kernels.py gpurun_out/pmc_x6/$b$j 300 2>/dev/null | head -3; done; done
