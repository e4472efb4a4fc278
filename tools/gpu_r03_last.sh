#!/bin/bash
# The closing measurement session of round 3 (one gpurun call, ~10 GPU-minutes): the suite, the bench lines, rocprof +
# PMC profiles at m = 10k and 100k, the BASELINE configurations, the row-view probe. usage: tools/gpu_r03_last.sh <commit>
COMMIT=${1:-unknown}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
out=gpurun_out/r03h; mkdir -p $out
export TMPDIR=/tmp
timeout 480 python -m pytest tests -m gpu -q --durations=10 > $out/gpu_tests.txt 2>&1; echo "gpu tests rc=$?"; grep -n "passed\|failed" $out/gpu_tests.txt | tail -2; grep -n "^FAILED\|^ERROR" $out/gpu_tests.txt | head
timeout 200 python bench.py --steps 20 --warmup 3 > $out/bench.log 2>$out/bench.err; echo "bench rc=$?"; grep '^{"metric"' $out/bench.log | cut -c1-300
timeout 100 python bench.py --steps 20 --warmup 3 --storage csc64 --no-cpu-baseline --probe-m 0 > $out/bench_csc64.log 2>&1; echo "bench csc64 rc=$?"
timeout 420 bash tools/gpu_prof_r03.sh r03h_prof $COMMIT > $out/prof_stdout.txt 2>&1; echo "prof rc=$?"
cp gpurun_out/r03h_prof/*.txt gpurun_out/r03h_prof/*.json $out/ 2>/dev/null
for m in 10000 100000; do grep '^{"metric"' gpurun_out/r03h_prof/trace_m$m.log | tail -1 > $out/bench_under_rocprof_m$m.jsonl; done
rm -rf gpurun_out/r03h_prof
timeout 150 python tools/run_configs.py --configs bunny,1k,pn5k,10k,30k,100k,300k --storage csc --reps 3 --no-cpu > $out/configs.jsonl 2>&1; echo "configs rc=$?"
timeout 120 python tools/rowview_probe.py --m 10000 30000 100000 300000 --reps 3 --profile > $out/rowview_probe.jsonl 2>&1; echo "probe rc=$?"
python - <<'PY'
import json
for l in open("gpurun_out/r03h/configs.jsonl"):
    if l.startswith("{"):
        d=json.loads(l); print(d["config"], "aff", d["gpu_affinity_ms"], "solve", d["gpu_solve_ms"], "passes", d["passes"], "on view", d["passes_on_view"], "rows", d["view_rows"], "pass us", d["gemv_us"], "view pass us", d["view_pass_us"])
PY
