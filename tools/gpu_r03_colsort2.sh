#!/bin/bash
# Round 3, fifth GPU session (short): a view's column order with the per-column layout of the sums
# (CLIPPER_HIP_RV_COLSORT=1) against none (default), and the measurement mode 2 (wrong sums: the order
# without any scattered store).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out
CLIPPER_HIP_RV_COLSORT=1 timeout 150 python -m pytest tests/test_gpu_rowview.py -x -q > $O/r03e_rowview_tests_colsort.txt 2>&1
echo "pytest (colsort=1) rc $?"; tail -2 $O/r03e_rowview_tests_colsort.txt
summ() {
python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: r = json.loads(l)
        except Exception: continue
        on = r["on"]
        print(f.split("/")[-1][:28], r["m"], "solve", on["solve_ms"], "passes", on["passes"], "view passes", on["view_passes"], "rows", on["rows"],
              "view bytes", on["view_bytes"], "build_ms", on["build_ms"], "pass_us", on["pass_us"], "view_pass_us", on["view_pass_us"], "hashes", on["u_hashes"], "nodes", on["nodes_sha"])
PY
}
for cfg in "off:CLIPPER_HIP_RV_COLSORT=0" "on:CLIPPER_HIP_RV_COLSORT=1"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 200 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 > $O/r03e_probe_$name.jsonl 2> $O/r03e_probe_$name.err
  env $envs timeout 100 python tools/rowview_probe.py --m 10000 100000 --reps 3 --profile > $O/r03e_probe_prof_$name.jsonl 2>> $O/r03e_probe_$name.err
  echo "== colsort $name"; summ $O/r03e_probe_$name.jsonl $O/r03e_probe_prof_$name.jsonl
done
CLIPPER_HIP_RV_COLSORT=2 timeout 60 python tools/rowview_probe.py --m 100000 --reps 1 --profile > $O/r03e_probe_prof_timingonly.jsonl 2> $O/r03e_probe_timingonly.err
echo "== colsort 2 (wrong sums, timing only) rc $?"; summ $O/r03e_probe_prof_timingonly.jsonl
