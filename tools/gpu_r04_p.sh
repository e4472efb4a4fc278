#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04p
for i in 1 2; do
  timeout 200 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --probe-m 0 > gpurun_out/r04p/bench_$i.log 2>&1
  grep -o '"value": [0-9.]*\|"solve_ms": [0-9.]*\|"affinity_ms": [0-9.]*' gpurun_out/r04p/bench_$i.log | head -3 | tr '\n' ' '; echo
done
timeout 300 python tools/run_configs.py --storage csc --configs 1k,pn5k,10k,30k --reps 5 --no-cpu > gpurun_out/r04p/configs.jsonl 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r04p/configs.jsonl'):
    if not line.startswith('{'): continue
    d=json.loads(line)
    print({k:d[k] for k in d if k in ('config','gpu_affinity_ms','gpu_solve_ms','passes','passes_on_view','views_built')})
PY
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile 2>&1 >/dev/null | grep "solve\]" | tail -3
