#!/bin/bash
# Profiles at a given commit (rounds 4, 5): rocprofv3 kernel stats of bench.py at m = 10k and 100k (default storage; views
# streamed as well at 10k; the dense fp32 store at 10k — north_star's literal row-blocked M*u), then the PMC counter
# passes (HBM traffic; LDS / VALU activity) of the same commands, tied to the sha256 of the kernel sources.
#   usage: tools/gpu_prof.sh <tag> <commit> [sizes]      -> gpurun_out/<tag>/ (kernel_stats_*.txt, pmc_*.txt, pmc_r06.json)
TAG=${1:-r04p}; COMMIT=${2:-unknown}; SIZES=${3:-"10000 100000"}
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
SHA=$(python bench.py --sources-sha)
prof() {  # <name> <env...> -- <bench args>
  local name=$1; shift
  local envs=(); while [ "$1" != "--" ]; do envs+=("$1"); shift; done; shift
  ( cd /tmp && env "${envs[@]}" timeout 240 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_$name -o trace -- python $ROOT/bench.py "$@" > $ROOT/$OUT/trace_$name.log 2>&1 )
  local DB=$(find $OUT/trace_$name -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats_$name.json > $OUT/kernel_stats_$name.txt 2>&1
}
for m in $SIZES; do
  steps=6; [ $m -ge 50000 ] && steps=2
  prof m$m -- --m $m --steps $steps --warmup 1 --no-cpu-baseline --probe-m 0
  [ $m -le 20000 ] && prof m${m}_views_streamed CLIPPER_HIP_VIEW_RESIDENT=0 -- --m $m --steps $steps --warmup 1 --no-cpu-baseline --probe-m 0
  prof m${m}_views_off CLIPPER_HIP_ROW_VIEW=0 -- --m $m --steps $steps --warmup 1 --no-cpu-baseline --probe-m 0
  [ $m -le 20000 ] && prof m${m}_dense_f32 -- --m $m --steps $steps --warmup 1 --no-cpu-baseline --probe-m 0 --storage f32
  bytes=$(grep '^{"metric"' $OUT/trace_m$m.log | tail -1 | python -c "import sys,json; print(json.loads(sys.stdin.readline())['roofline']['bytes_per_launch'])")
  B2="python $ROOT/bench.py --m $m --steps 2 --warmup 1 --no-cpu-baseline --probe-m 0 --no-profile"
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT" "SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32"; do
    name=$(echo $set | tr ' ' '_')
    ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $set -d $ROOT/$OUT/pmc_m${m}_$name -o pmc -- $B2 > $ROOT/$OUT/pmc_m${m}_$name.log 2>&1 )
  done
  python tools/pmc_summary.py --key m${m}_csc --bytes $bytes --commit $COMMIT --sources-sha $SHA --json $OUT/pmc_r06.json $(find $OUT -path "*pmc_m${m}_*" -name '*.db') > $OUT/pmc_m$m.txt 2>&1
done
find $OUT -name '*.db' -delete
for m in $SIZES; do echo "== m=$m"; head -12 $OUT/kernel_stats_m$m.txt; grep -E "k_gemv_slices|k_affinity_sym|k_tail|k_solve_view" $OUT/pmc_m$m.txt | head -40; done
