"""The REAL multi-process driver with two ranks on one GPU: two OS processes
open device 0 through clipper_hip_create_rank(..., rank, 2), each owns half of the columns of M
(its own fill, its own slices, its own passes) and they exchange the per-pass block through
clipper_hip_comm_init_callback with a gloo all-gather — the code path `bench.py --gpus N` runs
with RCCL, minus RCCL. Asserted: both ranks return the single-GPU node list bit for bit, the
same u on both ranks, and the solve terminates (the batched stop protocol keeps the ranks'
iteration counts equal past convergence; its CPU dry test is tests/test_batch_protocol.py)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, m, storage, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch
    import torch.distributed as tdist

    from clipper_amd import _abi as abi
    from clipper_amd import dist, synth

    dist.init_process_group("gloo")
    calls = [0]

    def allgather(block):
        calls[0] += 1
        t = torch.from_numpy(block)
        out = [torch.empty_like(t) for _ in range(world)]
        tdist.all_gather(out, t)
        return np.concatenate([o.numpy() for o in out])

    p = synth.make_euclidean_problem(m, 0.9, seed=77)       # identical on every rank
    g = abi.HipClipper(device=0, storage=storage, rank=rank, world=world)
    g.comm_init_callback(allgather)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    used = g.storage_in_use
    s = g.solve(p.u0)
    s2 = g.solve(p.u0)                                       # and once more on the same context
    assert s2.nodes.tolist() == s.nodes.tolist() and np.array_equal(s2.u, s.u)
    vs = g.view_stats()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.u, score=s.score, nodes=s.nodes,
             passes=s.n_passes, trials=s.n_trials, calls=calls[0], storage=used, window=g.window,
             views=vs.builds, view_passes=vs.view_passes)
    g.close()
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("m,storage_name", [(1500, "F32_CSC"), (6500, "F32_CSC"), (6500, "F32"), (1500, "F64_CSC")])
def test_two_processes_share_one_gpu(tmp_path, m, storage_name):
    # stdlib multiprocessing, not torch.multiprocessing: importing torch HERE, after libclipper_hip.so
    # has loaded the system HIP runtime, would put a second HIP runtime (torch's bundled one) into
    # this process (runtime notes); the workers import torch first, as documented
    import multiprocessing as mp

    from clipper_amd import _abi as abi
    from clipper_amd import synth

    storage = getattr(abi, "STORE_" + storage_name)
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, storage, str(tmp_path))) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=240)
    for pr in procs:
        if pr.is_alive():
            pr.terminate()
            pytest.fail("a rank did not terminate")
        assert pr.exitcode == 0
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert np.array_equal(r0["u"], r1["u"]) and r0["score"] == r1["score"]
    assert np.array_equal(r0["nodes"], r1["nodes"])
    assert r0["calls"] == r1["calls"] and r0["calls"] >= 2 * int(r0["passes"])   # two solves, one exchange per iteration
    assert int(r0["storage"]) == storage
    if m >= 3000 and storage_name.endswith("CSC"):   # both ranks went on hold together and built their views
        assert int(r0["views"]) >= 1 and int(r0["views"]) == int(r1["views"])
        assert int(r0["view_passes"]) == int(r1["view_passes"]) > 0
    # the single-GPU answer
    p = synth.make_euclidean_problem(m, 0.9, seed=77)
    one = abi.HipClipper(device=0, storage=storage)
    one.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = one.solve(p.u0)
    assert r0["nodes"].tolist() == s.nodes.tolist()
    assert abs(float(r0["score"]) - s.score) <= 1e-9 * abs(s.score)
    assert int(r0["trials"]) == s.n_trials
