#!/bin/bash
# Round 6, the closing session at a given commit, one gpurun call:
#   gpurun --timeout 3000 -- 'bash tools/gpu_r06.sh <commit>'            -> gpurun_out/r06z/
# rocprof kernel stats + PMC counters at m = 10k / 100k (-> pmc_r06.json, tied to the sha256 of the kernel sources), the
# fill kernel's stall counters, the configuration table, bench lines in five modes, the live sub-problem probe, host
# timing, the GPU suite.
COMMIT=${1:-unknown}
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/r06z; mkdir -p $OUT
export TMPDIR=/tmp
ROOT=$PWD
bash tools/gpu_prof.sh r06z $COMMIT "10000 100000" > $OUT/prof_session.txt 2>&1
# what k_affinity_sym waits for: wave counts, wait buckets, scalar / LDS / memory instruction mix
for m in 10000 100000; do
  B2="python $ROOT/bench.py --m $m --steps 2 --warmup 1 --no-cpu-baseline --probe-m 0 --no-profile"
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" "SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU" "SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_LDS_ADDR_CONFLICT"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-60)
    ( cd /tmp && timeout 240 rocprofv3 --kernel-trace --pmc $set -d $ROOT/$OUT/aff_m${m}_$name -o pmc -- $B2 > $ROOT/$OUT/aff_pmc_m${m}_$name.log 2>&1 )
  done
  python tools/pmc_summary.py $(find $OUT -path "*aff_m${m}_*" -name '*.db') 2>&1 | grep -E "k_affinity_sym|k_gemv_slices|k_solve_view" > $OUT/affinity_stall_counters_m$m.txt
done
find $OUT -name '*.db' -delete
timeout 600 python tools/run_configs.py --storage csc --configs bunny,1k,pn5k,10k --reps 5 > $OUT/configs_small.jsonl 2>$OUT/configs_small.err
timeout 600 python tools/run_configs.py --storage csc --configs 30k,100k,300k --reps 3 --no-cpu > $OUT/configs_large.jsonl 2>$OUT/configs_large.err
timeout 600 python tools/run_configs.py --storage csc64 --configs 10k,30k,100k,300k --reps 3 --no-cpu > $OUT/configs_csc64.jsonl 2>$OUT/configs_csc64.err
timeout 300 python bench.py > $OUT/bench.log 2> $OUT/bench.err
timeout 300 python bench.py --storage csc64 --no-cpu-baseline --probe-m 0 > $OUT/bench_csc64.log 2>&1
timeout 300 python bench.py --storage f32 --no-cpu-baseline --probe-m 0 > $OUT/bench_dense_f32.log 2>&1
CLIPPER_HIP_VIEW_RESIDENT=0 timeout 300 python bench.py --no-cpu-baseline --probe-m 0 > $OUT/bench_views_streamed.log 2>&1
CLIPPER_HIP_SUBPROBLEM=0 timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_sub_off.log 2>&1
CLIPPER_HIP_ROW_VIEW=0 timeout 300 python bench.py --no-cpu-baseline --probe-m 0 > $OUT/bench_views_off.log 2>&1
timeout 600 python tools/subproblem_probe.py --m 20000 30000 100000 300000 --reps 3 --modes noviews,views,sub --profile > $OUT/sub_probe.jsonl 2> $OUT/sub_probe.err
timeout 400 python tools/subproblem_probe.py --m 30000 100000 --storage csc64 --reps 3 --modes views,sub --profile > $OUT/sub_probe_csc64.jsonl 2>> $OUT/sub_probe.err
CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > /dev/null 2> $OUT/host_timing.txt
CLIPPER_HIP_HOST_TIMING=1 timeout 200 python bench.py --m 100000 --steps 2 --warmup 1 --no-cpu-baseline --probe-m 0 --no-profile > /dev/null 2> $OUT/host_timing_m100000.txt
CLIPPER_HIP_STAMPS=1 timeout 120 python tools/rvr_timeline.py > $OUT/rvr_timeline.txt 2>&1
timeout 1800 python -m pytest tests -m gpu -q > $OUT/gpu_tests_at_head.txt 2>&1; echo "suite rc=$?" > $OUT/summary.txt
tail -3 $OUT/gpu_tests_at_head.txt
grep -o '"value": [0-9.]*' $OUT/bench*.log | head
tail -30 $OUT/prof_session.txt
