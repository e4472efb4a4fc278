#!/bin/bash
# round 4, closing session at a given commit: profiles (rocprof + PMC), the configuration table, bench lines, timelines
COMMIT=${1:-unknown}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r04z; mkdir -p $OUT
bash tools/gpu_prof_r04.sh r04z $COMMIT "10000 100000" > $OUT/prof_session.txt 2>&1
timeout 600 python tools/run_configs.py --storage csc --configs bunny,1k,pn5k,10k --reps 5 > $OUT/configs_small.jsonl 2>$OUT/configs_small.err
timeout 600 python tools/run_configs.py --storage csc --configs 30k,100k,300k --reps 3 --no-cpu > $OUT/configs_large.jsonl 2>$OUT/configs_large.err
timeout 600 python tools/run_configs.py --storage csc64 --configs 10k,30k,100k,300k --reps 3 --no-cpu > $OUT/configs_csc64.jsonl 2>$OUT/configs_csc64.err
timeout 300 python bench.py > $OUT/bench.log 2> $OUT/bench.err
timeout 300 python bench.py --storage csc64 --no-cpu-baseline --probe-m 0 > $OUT/bench_csc64.log 2>&1
timeout 300 python bench.py --storage f32 --no-cpu-baseline --probe-m 0 > $OUT/bench_dense_f32.log 2>&1
CLIPPER_HIP_VIEW_RESIDENT=0 timeout 300 python bench.py --no-cpu-baseline --probe-m 0 > $OUT/bench_views_streamed.log 2>&1
CLIPPER_HIP_ROW_VIEW=0 timeout 300 python bench.py --no-cpu-baseline --probe-m 0 > $OUT/bench_views_off.log 2>&1
CLIPPER_HIP_STAMPS=1 timeout 120 python tools/rvr_timeline.py > $OUT/rvr_timeline.txt 2>&1
CLIPPER_HIP_STAMPS=1 timeout 120 python tools/rvr_timeline.py --storage csc64 >> $OUT/rvr_timeline.txt 2>&1
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > /dev/null 2> $OUT/host_timing.txt
timeout 400 python tools/rowview_probe.py --m 10000 30000 100000 300000 --profile > $OUT/rowview_probe.jsonl 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests_at_head.txt 2>&1; echo "suite rc=$?" > $OUT/summary.txt
tail -3 $OUT/gpu_tests_at_head.txt
grep -o '"value": [0-9.]*' $OUT/bench*.log | head
tail -30 $OUT/prof_session.txt
