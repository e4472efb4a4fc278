#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/r04h
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_rowview.py -x -q > gpurun_out/r04h/tests.txt 2>&1
echo "tests rc=$?" | tee gpurun_out/r04h/summary.txt
tail -5 gpurun_out/r04h/tests.txt
for pg in 1 0; do
  CLIPPER_HIP_PERSISTENT_GRID=$pg timeout 400 python tools/rowview_probe.py --m 30000 100000 300000 --profile > gpurun_out/r04h/probe_pg$pg.jsonl 2> gpurun_out/r04h/probe_pg$pg.err
  echo "persistent=$pg"; cat gpurun_out/r04h/probe_pg$pg.jsonl | cut -c1-600
done
