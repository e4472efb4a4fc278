#!/bin/bash
tag=${1:-r03i}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
timeout 100 python tools/rowview_probe.py --m 10000 30000 --reps 5 --profile > $out/probe.jsonl 2>&1
python - <<'PY'
import json
for l in open('gpurun_out/'+'TAG'+'/probe.jsonl'.replace('TAG','')) if False else open('gpurun_out/%s/probe.jsonl' % __import__('os').environ.get('TAGX','r03i')):
    if l.startswith('{'):
        d=json.loads(l); print(d['m'], 'off', d['off']['solve_ms'], d['off']['pass_us'], 'on', d['on']['solve_ms'], d['on']['pass_us'], d['on']['view_pass_us'], 'builds', d['on']['builds'], d['on']['build_ms'], 'vp', d['on']['view_passes'], d['on']['passes'], 'x', d['speedup'])
PY
timeout 60 python tools/pass_timeline.py 10000 > $out/timeline_10k.txt 2>&1; head -14 $out/timeline_10k.txt
timeout 60 python tools/pass_timeline.py 30000 > $out/timeline_30k.txt 2>&1; head -14 $out/timeline_30k.txt
