#!/bin/bash
out=gpurun_out/r03k; mkdir -p $out
export TMPDIR=/tmp
timeout 200 python -m pytest tests/test_gpu_rowview.py tests/test_gpu_configs.py -q > $out/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $out/tests.txt
TAGX=r03k timeout 120 python tools/rowview_probe.py --m 10000 30000 100000 --reps 3 --profile > $out/probe.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/r03k/probe.jsonl'):
    if l.startswith('{'):
        d=json.loads(l); print(d['m'], 'off', d['off']['solve_ms'], d['off']['pass_us'], 'on', d['on']['solve_ms'], d['on']['pass_us'], d['on']['view_pass_us'], 'builds', d['on']['builds'], d['on']['build_ms'], 'rows', d['on']['rows'], 'vp', d['on']['view_passes'], d['on']['passes'], 'x', d['speedup'])
PY
timeout 900 bash tools/gpu_prof_r03.sh r03k_prof 71aa3af
