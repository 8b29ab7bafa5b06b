#!/bin/bash
# tools/profile_bench.sh <tag>  -- run on the GPU box via gpurun: kernel-trace stats + PMC passes of bench.py.
# Outputs under gpurun_out/prof_<tag>/ ; copy the summaries you want judged into profiles/.
set -o pipefail
TAG=${1:-r1}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- $BENCH > $OUT/trace.log 2>&1 || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA" "GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  timeout -s KILL 300 rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -- $BENCH > $OUT/pmc_$name.log 2>&1 || { echo "pmc pass $pass failed"; tail -5 $OUT/pmc_$name.log; exit 1; }
done
find $OUT -name "*.csv" | head -50
