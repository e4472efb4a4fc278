#!/bin/bash
# One gpurun call: GPU parity suite, bench line, rocprofv3 kernel trace of the bench.
# usage: tools/gpu_session.sh <tag>      outputs under gpurun_out/<tag>/
set -u
TAG=${1:-run}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
( timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > $OUT/pytest.log
( timeout 300 python bench.py --steps 10 --warmup 2 2>&1 | tail -3 ) > $OUT/bench.log
ROOT=$PWD
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/prof -o trace -- python $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $ROOT/$OUT/prof.log 2>&1 )
DB=$(find $OUT/prof -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats.json > $OUT/kernel_stats.txt 2>&1
[ -n "$DB" ] && python tools/rocpd_timeline.py $DB > $OUT/timeline.txt 2>&1
find $OUT/prof -name '*.db' -size +20M -delete
tail -5 $OUT/pytest.log; cat $OUT/bench.log; cat $OUT/kernel_stats.txt
