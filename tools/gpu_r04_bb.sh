#!/bin/bash
# round 4, session bb: rocprof kernel stats of the small BASELINE configurations (cfg2 m = 1000, cfg4 PointNormal m = 5000)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04bb; mkdir -p $O
for c in 1k pn5k; do
  ( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace_$c -o trace -- python $GRAFT_REPO_ROOT/tools/run_configs.py --storage csc --configs $c --reps 8 --no-cpu > $GRAFT_REPO_ROOT/$O/run_$c.log 2>&1 )
  DB=$(find $O/trace_$c -name '*.db' | head -1)
  [ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $O/kernel_stats_$c.json > $O/kernel_stats_$c.txt 2>&1
  rm -rf $O/trace_$c
  head -14 $O/kernel_stats_$c.txt
done
