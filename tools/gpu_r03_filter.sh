#!/bin/bash
# Round 3, sixth GPU session: views filtered from M's slices (default) against views scored again from
# the points (CLIPPER_HIP_RV_BUILD=rect); the suite; the bench line.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out
timeout 480 python -m pytest tests -m gpu -x -q > $O/r03f_gpu_tests.txt 2>&1
echo "pytest rc $?" >> $O/r03f_gpu_tests.txt
tail -4 $O/r03f_gpu_tests.txt
summ() {
python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: r = json.loads(l)
        except Exception: continue
        on = r["on"]
        print(f.split("/")[-1][:28], r["m"], "solve", on["solve_ms"], "passes", on["passes"], "view passes", on["view_passes"], "builds", on["builds"], "rows", on["rows"],
              "view bytes", on["view_bytes"], "build_ms", on["build_ms"], "pass_us", on["pass_us"], "view_pass_us", on["view_pass_us"], "hashes", on["u_hashes"], "nodes", on["nodes_sha"])
PY
}
for cfg in "filter:CLIPPER_HIP_RV_BUILD=filter" "rect:CLIPPER_HIP_RV_BUILD=rect"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 200 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 > $O/r03f_probe_$name.jsonl 2> $O/r03f_probe_$name.err
  echo "== views built by $name"; summ $O/r03f_probe_$name.jsonl
done
timeout 200 python bench.py > $O/r03f_bench.log 2> $O/r03f_bench.err; grep '^{"metric"' $O/r03f_bench.log | cut -c1-300
