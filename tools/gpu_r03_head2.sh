#!/bin/bash
# at the round's HEAD: the bench line once more, bench under torch.distributed.run with one rank, the preflight with one RCCL rank
# and with two ranks on the one GPU through the callback
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 120 python bench.py > gpurun_out/r03o_bench.log 2> gpurun_out/r03o_bench.err; echo "bench rc=$?"; grep '^{"metric"' gpurun_out/r03o_bench.log | cut -c1-200
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --warmup 1 --probe-m 0 --no-cpu-baseline > gpurun_out/r03o_bench_torchrun1.log 2>&1; echo "bench under torchrun rc=$?"; grep '^{"metric"' gpurun_out/r03o_bench_torchrun1.log | cut -c1-160
timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 tools/multigpu_preflight.py --size 20000 > gpurun_out/r03o_preflight_rccl_1rank.txt 2>&1; echo "preflight rccl rc=$?"; grep -c PASS gpurun_out/r03o_preflight_rccl_1rank.txt; grep FAIL gpurun_out/r03o_preflight_rccl_1rank.txt | head -3
timeout 150 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29535 tools/multigpu_preflight.py --size 20000 --exchange callback --same-device > gpurun_out/r03o_preflight_callback_2ranks.txt 2>&1; echo "preflight callback rc=$?"; grep -c PASS gpurun_out/r03o_preflight_callback_2ranks.txt; grep FAIL gpurun_out/r03o_preflight_callback_2ranks.txt | head -3
