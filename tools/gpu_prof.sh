#!/bin/bash
# rocprofv3 kernel trace of bench.py. usage: tools/gpu_prof.sh <tag> [bench args...]; env passes through
TAG=${1:-prof}; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o trace -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline "$@" > $ROOT/$OUT/prof.log 2>&1 )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats.json > $OUT/kernel_stats.txt 2>&1
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB > $OUT/timeline.txt 2>&1
find $OUT/prof -name '*.db' -size +20M -delete
grep '"metric"' $OUT/prof.log | cut -c1-400; cat $OUT/kernel_stats.txt; cat $OUT/timeline.txt
