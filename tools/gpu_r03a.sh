#!/bin/bash
# first GPU session of round 3: the row view and the rectangular fill
tag=${1:-r03a}
out=gpurun_out/$tag
mkdir -p $out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build()" > $out/build.log 2>&1
timeout 300 python -m pytest tests/test_gpu_rowview.py -x -q -s > $out/rowview_tests.txt 2>&1
echo "rowview tests rc=$?" | tee -a $out/summary.txt
timeout 120 python tools/rowview_probe.py --m 10000 --reps 5 --profile > $out/probe_10k.jsonl 2>&1
echo "probe10k rc=$?" | tee -a $out/summary.txt
timeout 500 python -m pytest tests -m gpu -x -q > $out/gpu_tests.txt 2>&1
echo "gpu tests rc=$?" | tee -a $out/summary.txt
tail -5 $out/gpu_tests.txt
timeout 200 python tools/rowview_probe.py --m 30000 100000 --reps 2 --profile > $out/probe_big.jsonl 2>&1
echo "probe big rc=$?" | tee -a $out/summary.txt
timeout 120 python bench.py --steps 20 --warmup 3 > $out/bench.log 2>&1
echo "bench rc=$?" | tee -a $out/summary.txt
tail -3 $out/rowview_tests.txt
cat $out/probe_10k.jsonl $out/probe_big.jsonl
tail -2 $out/bench.log
