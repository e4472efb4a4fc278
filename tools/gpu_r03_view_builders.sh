#!/bin/bash
# the filter build with steps and headers in flight (one slice per tile) against the rectangular fill
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_rowview.py -x -q -k "filtered or handed" > $O/r03i_tests.txt 2>&1
echo "pytest rc $?"; tail -3 $O/r03i_tests.txt | cut -c1-300
for cfg in "filter:CLIPPER_HIP_RV_BUILD=filter" "rect:CLIPPER_HIP_RV_BUILD=rect"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 200 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 > $O/r03i_probe_$name.jsonl 2> $O/r03i_probe_$name.err
  echo "== views built by $name"
  python - $O/r03i_probe_$name.jsonl <<'PY'
import json, sys
for l in open(sys.argv[1]):
    try: r = json.loads(l)
    except Exception: continue
    on = r["on"]
    print(r["m"], "solve", on["solve_ms"], "passes", on["passes"], "view passes", on["view_passes"], "builds", on["builds"], "rows", on["rows"], "view bytes", on["view_bytes"], "build_ms", on["build_ms"], "hashes", on["u_hashes"], "nodes", on["nodes_sha"])
PY
done
