#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04m
timeout 400 python tools/run_configs.py --storage csc64 --configs 10k,30k,100k,300k --reps 2 --no-cpu > gpurun_out/r04m/configs_csc64.jsonl 2>&1
python - <<'PY'
import json
for line in open('gpurun_out/r04m/configs_csc64.jsonl'):
    if not line.startswith('{'): continue
    d=json.loads(line)
    print({k:d[k] for k in d if k in ('config','gpu_affinity_ms','gpu_solve_ms','passes','passes_on_view','views_built','view_rows','view_build_ms','view_pass_us','gemv_us','trials')})
PY
