#!/bin/bash
# rocprof kernel stats of tools/resident_check.py. usage: tools/gpu_resident_prof.sh <tag> <sizes>
TAG=$1; SIZES=${2:-100,1000,2048}; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o trace -- python $ROOT/tools/resident_check.py --sizes $SIZES --reps 5 --no-cpu > $ROOT/$OUT/prof.log 2>&1 )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats.json > $OUT/kernel_stats.txt 2>&1
find $OUT/prof -name '*.db' -delete
grep "^{" $OUT/prof.log | cut -c1-900; cat $OUT/kernel_stats.txt
