#!/bin/bash
# Round-3 profiles: rocprofv3 kernel stats of the bench at m = 10k and 100k, PMC counter passes (HBM
# traffic; LDS / VALU activity) of the same commands. usage: tools/gpu_prof_r03.sh <tag> <commit> [sizes]
TAG=${1:-r03p}; COMMIT=${2:-unknown}; SIZES=${3:-"10000 100000"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
for m in $SIZES; do
  steps=5; [ $m -ge 50000 ] && steps=2
  B="python $ROOT/bench.py --m $m --steps $steps --warmup 1 --no-cpu-baseline --probe-m 0"
  ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_m$m -o trace -- $B > $ROOT/$OUT/trace_m$m.log 2>&1 )
  DB=$(find $OUT/trace_m$m -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats_m$m.json > $OUT/kernel_stats_m$m.txt 2>&1
  ( cd /tmp && CLIPPER_HIP_ROW_VIEW=0 timeout 200 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_noview_m$m -o trace -- $B > $ROOT/$OUT/trace_noview_m$m.log 2>&1 )
  DB=$(find $OUT/trace_noview_m$m -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats_noview_m$m.json > $OUT/kernel_stats_noview_m$m.txt 2>&1
  bytes=$(grep '^{"metric"' $OUT/trace_m$m.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['roofline']['bytes_per_launch'])")
  B2="python $ROOT/bench.py --m $m --steps 2 --warmup 1 --no-cpu-baseline --probe-m 0 --no-profile"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
    name=$(echo $set | tr ' ' '_')
    ( cd /tmp && timeout 200 rocprofv3 --kernel-trace --pmc $set -d $ROOT/$OUT/pmc_m${m}_$name -o pmc -- $B2 > $ROOT/$OUT/pmc_m${m}_$name.log 2>&1 )
  done
  python tools/pmc_summary.py --key m${m}_csc --bytes $bytes --commit $COMMIT --json $OUT/pmc_r03.json $(find $OUT -path "*pmc_m${m}_*" -name '*.db') > $OUT/pmc_m$m.txt 2>&1
done
find $OUT -name '*.db' -size +8M -delete
for m in $SIZES; do echo "== m=$m"; head -12 $OUT/kernel_stats_m$m.txt; grep -E "k_gemv_slices|k_affinity_sym|k_tail" $OUT/pmc_m$m.txt | head -40; done
