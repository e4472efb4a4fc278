#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04q
for w in 0 1536 2304 3072 4608 6144; do
  if [ $w = 0 ]; then unset CLIPPER_HIP_CSC_WGS; else export CLIPPER_HIP_CSC_WGS=$w; fi
  timeout 200 python tools/rowview_probe.py --m 100000 --profile --reps 2 > gpurun_out/r04q/probe_$w.jsonl 2>/dev/null
  python - <<PY
import json
for line in open('gpurun_out/r04q/probe_$w.jsonl'):
    d=json.loads(line); on=d['on']; off=d['off']
    print('wgs=$w', 'off: solve %.1f pass %.1f | on: solve %.2f view_pass_us %.1f pass_us %.1f builds %d'%(off['solve_ms'],off['pass_us'],on['solve_ms'],on['view_pass_us'],on['pass_us'],on['builds']))
PY
done
