#!/bin/bash
# GPU suite + rocprof kernel stats of the bench with the compressed copy. usage: tools/gpu_csc_prof.sh <tag> [notest]
TAG=$1; OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
[ "${2:-}" != "notest" ] && ( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
ROOT=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o trace -- python $ROOT/bench.py --storage csc --steps 5 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/prof.log 2>&1 )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats.json > $OUT/kernel_stats.txt 2>&1
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB > $OUT/timeline.txt 2>&1
find $OUT/prof -name '*.db' -size +20M -delete
[ -f $OUT/pytest.log ] && tail -5 $OUT/pytest.log; tail -1 $OUT/prof.log; cat $OUT/kernel_stats.txt; head -40 $OUT/timeline.txt
