#!/bin/bash
# Round 3, fourth GPU session: the suite at HEAD; the view's column order and the tail's own fold, each
# against its switch (CLIPPER_HIP_RV_COLSORT=0, CLIPPER_HIP_FOLD=kernel); the bench line; DSD at 100k / 300k.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
O=gpurun_out
timeout 480 python -m pytest tests -m gpu -x -q > $O/r03d_gpu_tests.txt 2>&1
echo "pytest rc $?" >> $O/r03d_gpu_tests.txt
tail -3 $O/r03d_gpu_tests.txt
for cfg in "default:" "nocolsort:CLIPPER_HIP_RV_COLSORT=0" "foldkernel:CLIPPER_HIP_FOLD=kernel"; do
  name=${cfg%%:*}; envs=${cfg#*:}
  env $envs timeout 200 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 > $O/r03d_probe_$name.jsonl 2> $O/r03d_probe_$name.err
  env $envs timeout 120 python tools/rowview_probe.py --m 10000 100000 --reps 3 --profile > $O/r03d_probe_prof_$name.jsonl 2>> $O/r03d_probe_$name.err
  echo "== $name"; python - "$O/r03d_probe_$name.jsonl" "$O/r03d_probe_prof_$name.jsonl" <<'PY'
import json, sys
for f in sys.argv[1:]:
    for l in open(f):
        try: r = json.loads(l)
        except Exception: continue
        on = r["on"]
        print(f.split("/")[-1][:24], r["m"], "solve", on["solve_ms"], "passes", on["passes"], "view passes", on["view_passes"], "rows", on["rows"],
              "view bytes", on["view_bytes"], "build_ms", on["build_ms"], "pass_us", on["pass_us"], "view_pass_us", on["view_pass_us"], "hashes", on["u_hashes"], "nodes", on["nodes_sha"])
PY
done
timeout 200 python bench.py > $O/r03d_bench.log 2> $O/r03d_bench.err; grep '^{"metric"' $O/r03d_bench.log | cut -c1-400
CLIPPER_HIP_HOST_TIMING=1 timeout 200 python tools/dsd_timing.py --sizes 100000,300000 > $O/r03d_dsd_timing.jsonl 2> $O/r03d_dsd_timing.err
grep "^\[dsd\]" $O/r03d_dsd_timing.err; cat $O/r03d_dsd_timing.jsonl
