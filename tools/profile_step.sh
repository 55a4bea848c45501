#!/bin/bash
# Regenerates profiles/<tag>_*: a 6-step and a 2-step kernel trace (steady-state subtraction) + three PMC passes of bench.py.
# Usage (on the GPU box, from the repo root): bash tools/profile_step.sh <tag>
set -e
TAG=${1:-r02_x6}
R=$PWD
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# one stream: a kernel that shares the CUs with a kernel of another stream reads longer than it is (the untraced step overlaps
# the weight gradients and the second decoder with the main chain: ADM_SIDE_WGRAD / ADM_BRANCH_STREAM, on by default)
export ADM_SIDE_WGRAD=0 ADM_BRANCH_STREAM=0
rocprofv3 --kernel-trace --stats --output-format csv -d $O/long -- python3 $R/bench.py --steps 5 --warmup 1 --profile-only > $O/long.log 2>&1
echo "long trace done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/short -- python3 $R/bench.py --steps 1 --warmup 1 --profile-only > $O/short.log 2>&1
echo "short trace done"
for grp in FETCH_SIZE WRITE_SIZE "GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
  name=${grp%% *}
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$name -- python3 $R/bench.py --steps 1 --warmup 1 --profile-only > $O/pmc_$name.log 2>&1
  echo "pmc $name done"
done
cd $R
python tools/summarize_profiles.py $O/long $O/pmc_ $TAG 6 $O/short 2 > $O/summary.txt
head -30 $O/summary.txt
# the raw traces are large: keep the stats and counter CSVs only
find $O -name "*_kernel_trace.csv" -delete
