#!/bin/bash
# short check after the view-build switch: the row-view and multi-process tests, the probe at two sizes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_rowview.py tests/test_gpu_multiproc.py -x -q > gpurun_out/r03g_tests.txt 2>&1
echo "pytest rc $?"; tail -4 gpurun_out/r03g_tests.txt | cut -c1-300
timeout 100 python tools/rowview_probe.py --m 10000 100000 --reps 3 > gpurun_out/r03g_probe.jsonl 2> gpurun_out/r03g_probe.err
python - <<'PY'
import json
for l in open("gpurun_out/r03g_probe.jsonl"):
    try: r = json.loads(l)
    except Exception: continue
    on = r["on"]; print(r["m"], "solve", on["solve_ms"], "builds", on["builds"], "build_ms", on["build_ms"], "hashes", on["u_hashes"])
PY
