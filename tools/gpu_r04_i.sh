#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04i
timeout 500 python tools/rowview_probe.py --m 10000 30000 100000 300000 --profile > gpurun_out/r04i/probe.jsonl 2> gpurun_out/r04i/probe.err
python - <<'PY'
import json
for line in open('gpurun_out/r04i/probe.jsonl'):
    d=json.loads(line); on=d['on']; off=d['off']
    print(d['m'], 'off: solve %.2f pass %.1f trials %d | on: solve %.2f passes %d view_passes %d builds %d rows %d build_ms %.2f view_pass_us %.1f trials %d'%(off['solve_ms'],off['pass_us'],off['trials'],on['solve_ms'],on['passes'],on['view_passes'],on['builds'],on['rows'],on['build_ms'],on['view_pass_us'],on['trials']))
PY
