#!/bin/bash
# LDS / VALU activity of the pass kernel from PMC counters (own rocprofv3 passes, kernel trace only).
# usage: tools/gpu_pmc_lds.sh <tag>
TAG=${1:-pmclds}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
for set in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo $set | tr ' ' '_')
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc $set -d $ROOT/$OUT/$name -o pmc -- python $ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/$name.log 2>&1 )
done
python tools/rocpd_pmc.py $(find $OUT -name '*.db') > $OUT/pmc_lds.txt 2>&1
find $OUT -name '*.db' -size +20M -delete
grep -E "k_gemv_slices|k_tail|k_affinity|\.db" $OUT/pmc_lds.txt
