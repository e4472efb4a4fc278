#!/bin/bash
# round 4, session A: the resident solver on a view (new), then the whole GPU suite and a bench line
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04a
export CLIPPER_HIP_RESIDENT_DEBUG=1
timeout 900 python -m pytest tests/test_gpu_rv_resident.py -x -q -s > gpurun_out/r04a/rv_resident.txt 2>&1
echo "rv_resident rc=$?" | tee -a gpurun_out/r04a/summary.txt
unset CLIPPER_HIP_RESIDENT_DEBUG
timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r04a/bench.log 2> gpurun_out/r04a/bench.err
echo "bench rc=$?" | tee -a gpurun_out/r04a/summary.txt
tail -c 1500 gpurun_out/r04a/bench.log
CLIPPER_HIP_VIEW_RESIDENT=0 timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --probe-m 0 > gpurun_out/r04a/bench_streamed_views.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/gpu_tests.txt 2>&1
echo "suite rc=$?" | tee -a gpurun_out/r04a/summary.txt
tail -5 gpurun_out/r04a/gpu_tests.txt
tail -30 gpurun_out/r04a/rv_resident.txt
