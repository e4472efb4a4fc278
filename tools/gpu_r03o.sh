#!/bin/bash
out=gpurun_out/r03o; mkdir -p $out
export TMPDIR=/tmp
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/multigpu_preflight.py --size 12000 > $out/preflight_rccl1.txt 2>&1; echo "preflight rccl 1 rank rc=$?"; grep "step\|PREFLIGHT" $out/preflight_rccl1.txt | cut -c1-400
timeout 160 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29534 tools/multigpu_preflight.py --size 12000 --exchange callback --same-device > $out/preflight_cb2.txt 2>&1; echo "preflight callback 2 ranks rc=$?"; grep "step\|PREFLIGHT" $out/preflight_cb2.txt | cut -c1-400
