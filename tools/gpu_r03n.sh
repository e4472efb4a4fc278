#!/bin/bash
out=gpurun_out/r03n; mkdir -p $out
export TMPDIR=/tmp
timeout 300 python -m pytest tests/test_gpu_rowview.py tests/test_gpu_multiproc.py -q -x > $out/tests.txt 2>&1; echo "tests rc=$?"; tail -4 $out/tests.txt
timeout 200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_csc.py -q -x > $out/tests2.txt 2>&1; echo "tests2 rc=$?"; tail -3 $out/tests2.txt
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/multigpu_preflight.py --size 12000 > $out/preflight_rccl1.txt 2>&1; echo "preflight rccl 1 rank rc=$?"; grep "step\|PREFLIGHT" $out/preflight_rccl1.txt
timeout 160 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/multigpu_preflight.py --size 12000 --exchange callback --same-device > $out/preflight_cb2.txt 2>&1; echo "preflight callback 2 ranks rc=$?"; grep "step\|PREFLIGHT" $out/preflight_cb2.txt
