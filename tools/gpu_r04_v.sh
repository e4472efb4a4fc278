#!/bin/bash
# round 4, session v: the published decision (SolveArgs::plan_pub): parity subset, A/B against
# CLIPPER_HIP_PLAN_PUBLISH=0 over work-list sizes at m = 30k / 100k, the full timeline of a view pass
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04v; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_rowview.py tests/test_gpu_rv_resident.py tests/test_gpu_parity.py -x -q > $O/tests.txt 2>&1; echo "tests rc=$?" > $O/summary.txt
for pub in 1 0; do for w in 0 4608 6144 9216; do
  CLIPPER_HIP_PLAN_PUBLISH=$pub CLIPPER_HIP_CSC_WGS=$w timeout 300 python tools/rowview_probe.py --m 30000 100000 --profile > $O/probe_pub${pub}_w$w.jsonl 2> $O/probe_pub${pub}_w$w.err
  echo "probe pub=$pub w=$w rc=$?" >> $O/summary.txt
done; done
timeout 300 python tools/pass_timeline_full.py 100000 1 > $O/full_view_m100000.txt 2>&1
CLIPPER_HIP_CSC_WGS=6144 timeout 300 python tools/pass_timeline_full.py 100000 1 > $O/full_view_m100000_w6144.txt 2>&1
timeout 300 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
cat $O/summary.txt; tail -3 $O/tests.txt
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04v/probe_*.jsonl')):
    for line in open(f):
        d=json.loads(line); o=d['on']; off=d['off']
        print(f.split('/')[-1], d['m'], 'on', o['solve_ms'], 'passM', o['pass_us'], 'view', o['view_pass_us'], 'passes', o['passes'], o['trials'], '| off', off['solve_ms'], off['pass_us'])
PY
head -12 $O/full_view_m100000.txt
