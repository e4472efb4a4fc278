#!/bin/bash
# round 6, a quick look at a commit: the bench line, the live sub-problem at 100k / 300k, one kernel trace at 100k.
#   gpurun -- 'bash tools/gpu_r06_quick.sh <tag>'      -> gpurun_out/<tag>/
TAG=${1:-r06q}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
timeout 300 python bench.py > $OUT/bench.log 2> $OUT/bench.err
timeout 500 python tools/subproblem_probe.py --m 30000 100000 300000 --reps 2 --modes views,sub --profile > $OUT/sub_probe.jsonl 2> $OUT/sub_probe.err
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_m100000 -o trace -- python $ROOT/bench.py --m 100000 --steps 2 --warmup 1 --no-cpu-baseline --probe-m 0 > $ROOT/$OUT/trace_m100000.log 2>&1 )
DB=$(find $OUT/trace_m100000 -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $OUT/kernel_stats_m100000.json > $OUT/kernel_stats_m100000.txt 2>&1
find $OUT -name '*.db' -delete
cat $OUT/bench.log | cut -c1-6000; tail -3 $OUT/bench.err; cut -c1-1800 $OUT/sub_probe.jsonl; head -30 $OUT/kernel_stats_m100000.txt
