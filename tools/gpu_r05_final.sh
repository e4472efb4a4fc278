#!/bin/bash
# round 5, closing session at a given commit: profiles (rocprof + PMC -> pmc_r05.json), the configuration table, bench
# lines, timelines, the dense store at m = 100 000 under rocprof, the GPU suite.   gpurun -- 'bash tools/gpu_r05_final.sh <commit>'
COMMIT=${1:-unknown}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r05z; mkdir -p $OUT
export TMPDIR=/tmp
bash tools/gpu_prof.sh r05z $COMMIT "10000 100000" > $OUT/prof_session.txt 2>&1
timeout 600 python tools/run_configs.py --storage csc --configs bunny,1k,pn5k,10k --reps 5 > $OUT/configs_small.jsonl 2>$OUT/configs_small.err
timeout 600 python tools/run_configs.py --storage csc --configs 30k,100k,300k --reps 3 --no-cpu > $OUT/configs_large.jsonl 2>$OUT/configs_large.err
timeout 600 python tools/run_configs.py --storage csc64 --configs 10k,30k,100k,300k --reps 3 --no-cpu > $OUT/configs_csc64.jsonl 2>$OUT/configs_csc64.err
timeout 300 python bench.py > $OUT/bench.log 2> $OUT/bench.err
timeout 300 python bench.py --storage csc64 --no-cpu-baseline --probe-m 0 > $OUT/bench_csc64.log 2>&1
timeout 300 python bench.py --storage f32 --no-cpu-baseline --probe-m 0 > $OUT/bench_dense_f32.log 2>&1
CLIPPER_HIP_VIEW_RESIDENT=0 timeout 300 python bench.py --no-cpu-baseline --probe-m 0 > $OUT/bench_views_streamed.log 2>&1
CLIPPER_HIP_ROW_VIEW=0 timeout 300 python bench.py --no-cpu-baseline --probe-m 0 > $OUT/bench_views_off.log 2>&1
# north_star's literal kernel in the HBM regime: the dense fp32 store at m = 100 000 (40 GB), one step under rocprof
ROOT=$PWD; ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats -d $ROOT/$OUT/trace_m100000_dense_f32 -o trace -- python $ROOT/bench.py --m 100000 --steps 1 --warmup 0 --storage f32 --no-cpu-baseline --probe-m 0 > $ROOT/$OUT/trace_m100000_dense_f32.log 2>&1 )
DB=$(find $OUT/trace_m100000_dense_f32 -name '*.db' | head -1); [ -n "$DB" ] && python tools/rocpd_stats.py $DB > $OUT/kernel_stats_m100000_dense_f32.txt 2>&1; find $OUT -name '*.db' -delete
CLIPPER_HIP_STAMPS=1 timeout 120 python tools/rvr_timeline.py > $OUT/rvr_timeline.txt 2>&1
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > /dev/null 2> $OUT/host_timing.txt
timeout 400 python tools/rowview_probe.py --m 10000 30000 100000 300000 --profile > $OUT/rowview_probe.jsonl 2>/dev/null
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_tests_at_head.txt 2>&1; echo "suite rc=$?" > $OUT/summary.txt
tail -3 $OUT/gpu_tests_at_head.txt
grep -o '"value": [0-9.]*' $OUT/bench*.log | head
tail -30 $OUT/prof_session.txt
