#!/bin/bash
# round 4, session y: k_scal_fold with its 32 loads in flight; the row list of a view in one launch (m <= 16 384)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04y; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/gpu_tests.txt 2>&1; echo "suite rc=$?" > $O/summary.txt
timeout 300 python tools/rowview_probe.py --m 10000 30000 100000 --profile > $O/probe.jsonl 2> $O/probe.err; echo "probe rc=$?" >> $O/summary.txt
timeout 300 python bench.py --no-cpu-baseline > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > /dev/null 2> $O/host_timing.txt
cat $O/summary.txt; tail -3 $O/gpu_tests.txt
python - <<'PY'
import json
for line in open('gpurun_out/r04y/probe.jsonl'):
    d=json.loads(line); o=d['on']; off=d['off']
    print(f"{d['m']:7d} on {o['solve_ms']:8.3f} passM {o['pass_us']:8.1f} view {o['view_pass_us']:7.1f} passes {o['passes']} {o['trials']} build {o['build_ms']} | off {off['solve_ms']:8.3f} {off['pass_us']:8.1f}")
d=json.loads(open('gpurun_out/r04y/bench.log').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['affinity_ms'], d['solve_ms'], d['roofline']['achieved'], d['row_view']['build_ms'], d['scaling_probe']['ms_per_step'], d['scaling_probe']['solve_ms'])
PY
grep '^\[solve\]' $O/host_timing.txt | tail -3; grep '^\[view\]' $O/host_timing.txt | tail -3
