#!/bin/bash
# round 5: ONE parameterised GPU session script (round 4 left 40 one-off launchers behind).
#   gpurun -- 'bash tools/gpu_r05.sh <session> <step> [<step> ...]'      outputs under gpurun_out/<session>/
# steps: suite | tests:<pytest args> | bench | bench:<args> | adj | hosttiming | rvrtimeline | prof:<m list> | cmd:<shell>
S=${1:-r05}; shift
cd "$GRAFT_REPO_ROOT" || exit 1
OUT=gpurun_out/$S; mkdir -p $OUT
for step in "$@"; do
  t0=$(date +%s)
  case "$step" in
    suite) timeout 1500 python -m pytest tests -m gpu -q -x > $OUT/suite.txt 2>&1; echo "suite rc=$?" >> $OUT/summary.txt; tail -5 $OUT/suite.txt ;;
    tests:*) timeout 1200 python -m pytest ${step#tests:} -q -x -s > $OUT/tests_$(echo "${step#tests:}" | tr -c 'a-zA-Z0-9\n' _ | cut -c1-60).txt 2>&1; echo "$step rc=$?" >> $OUT/summary.txt; tail -15 $OUT/tests_*.txt | tail -25 ;;
    bench) timeout 300 python bench.py > $OUT/bench.log 2> $OUT/bench.err; tail -1 $OUT/bench.log | cut -c1-600 ;;
    bench:*) n=$(echo "${step#bench:}" | tr -c 'a-zA-Z0-9\n' _ | cut -c1-50); timeout 600 python bench.py ${step#bench:} > $OUT/bench_$n.log 2> $OUT/bench_$n.err; tail -1 $OUT/bench_$n.log | cut -c1-600 ;;
    adj) timeout 900 python tools/maxiniters_adjudicate.py gpu > $OUT/adj_gpu.log 2>&1; echo "adj rc=$?" >> $OUT/summary.txt; tail -2 $OUT/adj_gpu.log | cut -c1-300 ;;
    hosttiming) CLIPPER_HIP_HOST_TIMING=1 timeout 120 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --probe-m 0 --no-profile > /dev/null 2> $OUT/host_timing.txt; grep "^\[solve\] init" $OUT/host_timing.txt | tail -3 ;;
    rvrtimeline) CLIPPER_HIP_STAMPS=1 timeout 120 python tools/rvr_timeline.py > $OUT/rvr_timeline.txt 2>&1; tail -20 $OUT/rvr_timeline.txt ;;
    prof:*) bash tools/gpu_prof.sh $S $(cat .commit_for_prof 2>/dev/null || echo unknown) "${step#prof:}" > $OUT/prof_session.txt 2>&1; tail -5 $OUT/prof_session.txt ;;
    cmd:*) bash -c "${step#cmd:}" > $OUT/cmd_$(date +%s).txt 2>&1; tail -30 $OUT/cmd_*.txt | tail -40 ;;
  esac
  echo "[$step] $(( $(date +%s) - t0 )) s" | tee -a $OUT/summary.txt
done
