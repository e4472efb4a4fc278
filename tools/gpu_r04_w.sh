#!/bin/bash
# round 4, session w: the work list of a long pass — cost of a chunk besides its steps (C0), two tiers of items, items per strip
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04w; mkdir -p $O
run() { # name, env...
  local name=$1; shift
  env "$@" timeout 300 python tools/rowview_probe.py --m 30000 100000 --profile > $O/probe_$name.jsonl 2> $O/probe_$name.err
  echo "probe $name rc=$?" >> $O/summary.txt
}
: > $O/summary.txt
run base X=1
run c0_4 CLIPPER_HIP_PLAN_C0=4
run c0_6 CLIPPER_HIP_PLAN_C0=6
run c0_10 CLIPPER_HIP_PLAN_C0=10
run big70_s2 CLIPPER_HIP_PLAN_BIG=0.7 CLIPPER_HIP_PLAN_SMALL=2
run big60_s2 CLIPPER_HIP_PLAN_BIG=0.6 CLIPPER_HIP_PLAN_SMALL=2
run big50_s2 CLIPPER_HIP_PLAN_BIG=0.5 CLIPPER_HIP_PLAN_SMALL=2
run big75_s3 CLIPPER_HIP_PLAN_BIG=0.75 CLIPPER_HIP_PLAN_SMALL=3
run big80_s2 CLIPPER_HIP_PLAN_BIG=0.8 CLIPPER_HIP_PLAN_SMALL=2
run per6 CLIPPER_HIP_PLAN_PER=6
run per12 CLIPPER_HIP_PLAN_PER=12
run big70_s2_c06 CLIPPER_HIP_PLAN_BIG=0.7 CLIPPER_HIP_PLAN_SMALL=2 CLIPPER_HIP_PLAN_C0=6
cat $O/summary.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04w/probe_*.jsonl')):
    for line in open(f):
        d=json.loads(line); o=d['on']; off=d['off']
        print(f"{f.split('/')[-1]:28s} {d['m']:7d} on {o['solve_ms']:8.3f} passM {o['pass_us']:8.1f} view {o['view_pass_us']:7.1f} passes {o['passes']} {o['trials']} | off {off['solve_ms']:8.3f} {off['pass_us']:8.1f}")
PY
