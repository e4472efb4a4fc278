#!/bin/bash
# round 4, session x: the tail without its fold (k_tail BIG: <= 64 rows, a thread walks its elements) — the whole
# GPU suite, then probes at 30k / 100k / 300k: 512 vs 256 threads, two tiers for M's work list
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04x; mkdir -p $O
: > $O/summary.txt
run() { local name=$1; local sizes=$2; shift; shift
  env "$@" timeout 600 python tools/rowview_probe.py --m $sizes --profile > $O/probe_$name.jsonl 2> $O/probe_$name.err
  echo "probe $name rc=$?" >> $O/summary.txt; }
run t512 "30000 100000 300000" X=1
run t256 "30000 100000 300000" CLIPPER_HIP_TAIL_BIG=256
run t512_tierM "30000 100000 300000" CLIPPER_HIP_PLAN_BIG=0.7
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "suite rc=$?" >> $O/summary.txt
( cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace100k -o trace -- python $GRAFT_REPO_ROOT/bench.py --m 100000 --steps 2 --warmup 1 --no-cpu-baseline --probe-m 0 > $GRAFT_REPO_ROOT/$O/trace100k.log 2>&1 )
DB=$(find $O/trace100k -name '*.db' | head -1)
[ -n "$DB" ] && python tools/rocpd_stats.py $DB --json $O/kernel_stats_m100000.json > $O/kernel_stats_m100000.txt 2>&1
rm -rf $O/trace100k
cat $O/summary.txt; tail -4 $O/gpu_tests.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04x/probe_*.jsonl')):
    for line in open(f):
        d=json.loads(line); o=d['on']; off=d['off']
        print(f"{f.split('/')[-1]:24s} {d['m']:7d} on {o['solve_ms']:8.3f} passM {o['pass_us']:8.1f} view {o['view_pass_us']:7.1f} passes {o['passes']} {o['trials']} | off {off['solve_ms']:8.3f} {off['pass_us']:8.1f} passes {off['passes']}")
PY
head -12 $O/kernel_stats_m100000.txt
