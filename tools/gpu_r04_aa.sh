#!/bin/bash
# round 4, session aa: staggered item sizes for lists of several rounds (views / M), probes at 30k / 100k / 300k
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04aa; mkdir -p $O
: > $O/summary.txt
run() { local name=$1; shift
  env "$@" timeout 600 python tools/rowview_probe.py --m 30000 100000 300000 --profile > $O/probe_$name.jsonl 2> $O/probe_$name.err
  echo "probe $name rc=$?" >> $O/summary.txt; }
run base X=1
run v30 CLIPPER_HIP_PLAN_STAGGER_VIEW=0.3
run v45 CLIPPER_HIP_PLAN_STAGGER_VIEW=0.45
run v45_m30 CLIPPER_HIP_PLAN_STAGGER_VIEW=0.45 CLIPPER_HIP_PLAN_STAGGER=0.3
run base2 X=2
cat $O/summary.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04aa/probe_*.jsonl')):
    for line in open(f):
        d=json.loads(line); o=d['on']; off=d['off']
        print(f"{f.split('/')[-1][6:-6]:10s} {d['m']:7d} on {o['solve_ms']:8.3f} passM {o['pass_us']:8.1f} view {o['view_pass_us']:7.1f} passes {o['passes']} {o['trials']} | off {off['solve_ms']:8.3f} {off['pass_us']:8.1f} passes {off['passes']}")
PY
