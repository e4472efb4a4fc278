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


def _worker(rank, world, port, m, storage, out_dir, rho=0.9, seed=77, env=None):
    sys.path.insert(0, ROOT)
    os.environ.update(env or {})
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

    pointnormal = bool((env or {}).get("TEST_POINTNORMAL"))
    p = (synth.make_pointnormal_problem(m, rho, seed=seed) if pointnormal
         else synth.make_euclidean_problem(m, rho, seed=seed))     # identical on every rank
    g = abi.HipClipper(device=0, storage=storage, rank=rank, world=world)
    g.comm_init_callback(allgather)
    if pointnormal:
        g.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
    else:
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    used = g.storage_in_use
    s = g.solve(p.u0)
    vs1 = g.view_stats()
    s2 = g.solve(p.u0)                                       # and once more on the same context
    assert s2.nodes.tolist() == s.nodes.tolist() and np.array_equal(s2.u, s.u)
    vs = g.view_stats()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.u, score=s.score, nodes=s.nodes,
             passes=s.n_passes, trials=s.n_trials, calls=calls[0], storage=used, window=g.window,
             views=vs.builds, view_passes=vs.view_passes, resident=vs1.resident_launches, giveups=vs1.resident_giveups,
             resident2=vs.resident_launches, giveups2=vs.resident_giveups, ifinal=s.ifinal)
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
    assert r0["calls"] == r1["calls"]
    if int(r0["resident"]) == 0:   # two solves, one exchange per iteration (the iterations inside a resident launch on the
        assert r0["calls"] >= 2 * int(r0["passes"])   # view's replica have none: round 5)
    assert int(r0["resident"]) == int(r1["resident"]) and int(r0["giveups"]) == int(r1["giveups"])
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


def _run_ranks(tmp_path, world, m, storage, rho, seed, env):
    import multiprocessing as mp
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, world, port, m, storage, str(tmp_path), rho, seed, env)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=240)
    for pr in procs:
        if pr.is_alive():
            pr.terminate()
            pytest.fail("a rank did not terminate")
        assert pr.exitcode == 0
    return [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]


@pytest.mark.parametrize("storage_name", ["F32_CSC", "F64_CSC"])
def test_two_ranks_run_the_resident_solver_on_a_replica_of_the_view(tmp_path, monkeypatch, storage_name):
    """with column shards the headline problem's 25 iterations on its view no longer stream with
    an exchange each — every rank builds a REPLICA of the view over all columns (4.8 MB, from the replicated points)
    and runs k_solve_view_resident on it redundantly: the same bits on every rank, no exchange inside the launch.
    Two OS processes on ONE GPU: 2 x 120 units fit the chip side by side (CLIPPER_HIP_VIEW_RESIDENT_WGS; the
    default, two thirds of the CUs per rank, would have the two launches wait for each other's CUs) — the one-GPU
    reference runs with the same number of units, so that u can be compared bit for bit."""
    from clipper_amd import _abi as abi
    from clipper_amd import synth
    from oracle import clipper_ref as ref

    storage = getattr(abi, "STORE_" + storage_name)
    env = {"CLIPPER_HIP_VIEW_RESIDENT_WGS": "120"}
    r0, r1 = _run_ranks(tmp_path, 2, 10000, storage, 0.95, 12345, env)
    assert int(r0["resident"]) == int(r1["resident"]) >= 1 and int(r0["giveups"]) == int(r1["giveups"]) == 0
    assert np.array_equal(r0["u"], r1["u"]) and r0["score"] == r1["score"] and np.array_equal(r0["nodes"], r1["nodes"])
    assert int(r0["view_passes"]) == int(r1["view_passes"]) > 0
    # (the iterations inside the launch have no exchange: the streamed route makes one per pass — 2 x 36 and the
    # no-op iterations of the stop protocol on top)
    assert int(r0["calls"]) == int(r1["calls"])
    print(f"exchanges of two solves: {int(r0['calls'])} for 2 x {int(r0['passes'])} passes, {int(r0['resident'])} resident launch(es) per solve")
    monkeypatch.setenv("CLIPPER_HIP_VIEW_RESIDENT_WGS", "120")
    p = synth.make_euclidean_problem(10000, 0.95, seed=12345)
    one = abi.HipClipper(device=0, storage=storage)
    one.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = one.solve(p.u0)
    assert one.view_stats().resident_launches == 1
    monkeypatch.delenv("CLIPPER_HIP_VIEW_RESIDENT_WGS")
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    sr = r.solve(p.u0)
    assert r0["nodes"].tolist() == s.nodes.tolist() == sr.nodes.tolist()
    assert int(r0["trials"]) == s.n_trials == sr.n_trials == 66 and int(r0["ifinal"]) == s.ifinal == sr.ifinal
    assert abs(float(r0["score"]) - s.score) <= 1e-10 * abs(s.score)
    # the passes on M before the view sum their partials per shard (other slots): u agrees to rounding, the
    # resident part adds nothing to that
    assert np.allclose(r0["u"], s.u, rtol=0, atol=1e-9)
    one.close()


def test_a_rank_whose_launch_gives_up_takes_the_other_rank_with_it(tmp_path):
    """One rank's resident launch times out (forced on rank 1 only): the ranks agree on the outcome through the
    exchange, rank 0 puts its state back as it was before its launch, and both stream the view's iterations — same
    iteration counts (no collective left hanging), same result as the streamed solve, both report the give-up."""
    import multiprocessing as mp

    from clipper_amd import _abi as abi
    from clipper_amd import synth

    port = _free_port()
    ctx = mp.get_context("spawn")
    env = {"CLIPPER_HIP_VIEW_RESIDENT_WGS": "120"}
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 10000, abi.STORE_F32_CSC, str(tmp_path), 0.95, 12345,
                                                dict(env, **({"CLIPPER_HIP_VIEW_RESIDENT_TIMEOUT_TICKS": "-1"} if r == 1 else {}))))
             for r in range(2)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(timeout=240)
    for pr in procs:
        if pr.is_alive():
            pr.terminate()
            pytest.fail("a rank did not terminate")
        assert pr.exitcode == 0
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(2))
    assert int(r0["resident"]) == int(r1["resident"]) == 0
    assert int(r0["giveups"]) >= 1 and int(r1["giveups"]) >= 1      # (rank 0's is "a peer gave up")
    assert int(r0["resident2"]) == int(r1["resident2"]) == 0 and int(r0["giveups2"]) == int(r1["giveups2"]) == 0   # backing off, together
    assert np.array_equal(r0["u"], r1["u"]) and np.array_equal(r0["nodes"], r1["nodes"]) and int(r0["calls"]) == int(r1["calls"])
    p = synth.make_euclidean_problem(10000, 0.95, seed=12345)
    one = abi.HipClipper(device=0, storage=abi.STORE_F32_CSC)
    one.set_row_view(2)
    one.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = one.solve(p.u0)
    assert r0["nodes"].tolist() == s.nodes.tolist() and int(r0["trials"]) == s.n_trials
    one.close()


def test_two_ranks_pointnormal_with_views(tmp_path):
    """PointNormalDistance (6-DoF surfels) through the multi-process driver with row views: the per-rank fills and —
    where the view is small enough — the replica are scored by the PointNormal variant of the rectangular kernel.
    Both ranks equal, and equal to one GPU's node list."""
    from clipper_amd import _abi as abi
    from clipper_amd import synth

    m, rho, seed = 9000, 0.93, 31
    # (these solves are short — ten trials: the build side of the view policy's cost model is scaled down so that a view
    # is asked for at all)
    env = {"CLIPPER_HIP_VIEW_RESIDENT_WGS": "120", "TEST_POINTNORMAL": "1", "CLIPPER_HIP_RV_BUILD_SCALE": "0.02"}
    r0, r1 = _run_ranks(tmp_path, 2, m, abi.STORE_F32_CSC, rho, seed, env)
    assert np.array_equal(r0["u"], r1["u"]) and np.array_equal(r0["nodes"], r1["nodes"]) and int(r0["calls"]) == int(r1["calls"])
    assert int(r0["resident"]) == int(r1["resident"]) >= 1 and int(r0["giveups"]) == int(r1["giveups"]) == 0
    assert int(r0["views"]) == int(r1["views"]) >= 1
    p = synth.make_pointnormal_problem(m, rho, seed=seed)
    one = abi.HipClipper(device=0, storage=abi.STORE_F32_CSC)
    one.score_pairwise_consistency_pointnormal(p.D1, p.D2, p.A)
    s = one.solve(p.u0)
    st = one.view_stats()
    print(f"PointNormal m={m}: views {int(r0['views'])} (one GPU: {st.builds}, {st.rows} rows), resident launches per rank "
          f"{int(r0['resident'])}, trials {int(r0['trials'])} / one GPU {s.n_trials}")
    assert sorted(r0["nodes"].tolist()) == sorted(s.nodes.tolist()) and int(r0["ifinal"]) == s.ifinal
    assert abs(float(r0["score"]) - s.score) <= 1e-9 * abs(s.score)
    one.close()
