#!/bin/bash
# the measurement session of round 3 (one gpurun call): bench lines, the BASELINE configurations, the row-view
# probe, rocprof + PMC profiles, the pass timelines, the GPU suite
out=gpurun_out/r03b; mkdir -p $out
export TMPDIR=/tmp
timeout 200 python bench.py --steps 20 --warmup 3 > $out/bench.log 2>$out/bench.err; echo "bench rc=$?"; tail -1 $out/bench.log | cut -c1-400
timeout 200 python bench.py --steps 20 --warmup 3 --storage csc64 --no-cpu-baseline > $out/bench_csc64.log 2>&1; echo "bench csc64 rc=$?"
CLIPPER_HIP_ROW_VIEW=0 timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline > $out/bench_views_off.log 2>&1; echo "bench views off rc=$?"
timeout 400 python tools/run_configs.py --configs bunny,1k,pn5k,10k,30k --storage csc --reps 5 > $out/configs.jsonl 2>&1; echo "configs rc=$?"
timeout 200 python tools/run_configs.py --configs 100k,300k --storage csc --reps 3 --no-cpu >> $out/configs.jsonl 2>&1; echo "configs big rc=$?"
timeout 300 python tools/run_configs.py --configs 10k,30k,100k,300k --storage csc64 --reps 3 --no-cpu > $out/configs_csc64.jsonl 2>&1; echo "configs csc64 rc=$?"
timeout 200 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 --profile > $out/rowview_probe.jsonl 2>&1; echo "probe rc=$?"
timeout 60 python tools/pass_timeline.py 10000 > $out/pass_timeline_m10000.txt 2>&1
CLIPPER_HIP_ROW_VIEW=0 timeout 60 python tools/pass_timeline.py 10000 > $out/pass_timeline_m10000_views_off.txt 2>&1
timeout 900 bash tools/gpu_prof_r03.sh r03b_prof df33747 > $out/prof_stdout.txt 2>&1; echo "prof rc=$?"
cp gpurun_out/r03b_prof/*.txt gpurun_out/r03b_prof/*.json $out/ 2>/dev/null
for m in 10000 100000; do grep '^{"metric"' gpurun_out/r03b_prof/trace_m$m.log | tail -1 > $out/bench_under_rocprof_m$m.jsonl; done
rm -rf gpurun_out/r03b_prof
timeout 600 python -m pytest tests -m gpu -q --durations=12 > $out/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed" $out/gpu_tests.txt | tail -2; grep -n "^FAILED\|^ERROR" $out/gpu_tests.txt | head
python - <<'PY'
import json
for f in ("configs.jsonl","configs_csc64.jsonl"):
    for l in open("gpurun_out/r03b/"+f):
        if l.startswith("{"):
            d=json.loads(l); print(f[:-6], d["config"], "aff", d["gpu_affinity_ms"], "solve", d["gpu_solve_ms"], "passes", d["passes"], "on view", d["passes_on_view"], "rows", d["view_rows"], "pass us", d["gemv_us"], "view pass us", d["view_pass_us"], "cpu", d.get("cpu_affinity_ms"), d.get("cpu_solve_ms"), "same", d.get("set_identical"), d.get("rel_dscore"))
PY
