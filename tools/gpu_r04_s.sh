#!/bin/bash
# round 4, session s: the marks hand-over (bench roofline with a resident launch), the units-count test, and
# the symmetric-half pass in the stand-alone harness (tools/slice_tune.hip, CLIPPER_SL_XMODE=4) beside mode 0
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r04s; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_rv_resident.py -x -q > $O/rv_resident.txt 2>&1; echo "rv_resident rc=$?" > $O/summary.txt
timeout 600 python bench.py > $O/bench.log 2> $O/bench.err; echo "bench rc=$?" >> $O/summary.txt
for x in 0 4; do
  timeout 300 tools/_bin/slice_tune_x$x 10000 0.105 0.05 200 0 1 > $O/sym_m10000_x$x.txt 2>&1; echo "x$x 10k rc=$?" >> $O/summary.txt
  timeout 900 tools/_bin/slice_tune_x$x 40000 0.105 0.05 20 0 1 > $O/sym_m40000_x$x.txt 2>&1; echo "x$x 40k rc=$?" >> $O/summary.txt
done
cat $O/summary.txt; tail -3 $O/rv_resident.txt; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r04s/bench.log').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['roofline']['achieved'], d['roofline']['frac'], d['gemv_avg_us'], d['row_view'])
PY
tail -25 $O/sym_m10000_x4.txt; tail -22 $O/sym_m40000_x4.txt; tail -20 $O/sym_m40000_x0.txt
