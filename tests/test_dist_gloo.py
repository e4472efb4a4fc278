"""N > 1 path on CPU: world_size-2 `gloo` process groups (no GPU needed).

What runs here is the HOST logic of the sharded path (clipper_amd/dist.py) and the sharding
protocol itself — column slices, the [P][2][W] gathered layout the kernels index with `ab_at`,
one all-gather per pass, every rank recomputing all O(m) work redundantly — with the oracle's
numpy loop standing in for the device kernels. The device side of the same protocol is
exercised with several logical shards on one GPU in tests/test_gpu_parity.py."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, m, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch
    import torch.distributed as tdist

    from clipper_amd import dist, synth
    from oracle import clipper_ref as ref

    d = dist.init_process_group("gloo")
    assert d.get_world_size() == world

    # --- bootstrap helpers used by bench.py ---------------------------------------------
    uid = bytes(range(128)) if rank == 0 else None
    got = dist.broadcast_bytes(uid, 128, src=0)
    assert got == bytes(range(128))
    assert dist.max_over_ranks(1.0 + rank) == float(world)

    # --- the sharded solve protocol --------------------------------------------------------
    p = synth.make_euclidean_problem(m, 0.9, seed=21)       # identical on every rank
    Mup, _ = ref.numpy_affinity_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    Ms = Mup + Mup.T
    Cs = (Ms != 0).astype(float)
    W = dist.shard_pitch(m, world)
    c0, c1 = dist.shard_columns(m, world, rank)
    npass = [0]

    def sharded_matvec(x):
        # what k_gemv + k_reduce produce on this rank: sums over all rows for the owned columns
        blk = dist.pack_block(Ms[:, c0:c1].T @ x, Cs[:, c0:c1].T @ x, W)
        gathered = [torch.zeros(2 * W, dtype=torch.float64) for _ in range(world)]
        tdist.all_gather(gathered, torch.from_numpy(blk))       # one exchange per pass
        npass[0] += 1
        ab = np.concatenate([g.numpy() for g in gathered])
        return dist.unpack_gathered(ab, m, world)

    s = ref.numpy_solve(None, None, p.u0, matvec=sharded_matvec)
    # replicated state must be bit-identical on every rank (no scalar all-reduce is needed)
    assert dist.all_equal_over_ranks(s.u)
    assert dist.all_equal_over_ranks(np.array([s.score, s.d]))
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), u=s.u, score=s.score, nodes=s.nodes,
             npass=npass[0])
    tdist.barrier()
    tdist.destroy_process_group()


@pytest.mark.parametrize("m", [300, 517])
def test_world_size_2_gloo_sharded_protocol(tmp_path, m):
    import torch.multiprocessing as mp

    from clipper_amd import synth
    from oracle import clipper_ref as ref

    world, port = 2, _free_port()
    mp.start_processes(_worker, args=(world, port, m, str(tmp_path)), nprocs=world, join=True,
                       start_method="spawn")
    r0, r1 = (np.load(tmp_path / f"rank{r}.npz") for r in range(world))
    assert np.array_equal(r0["u"], r1["u"]) and r0["score"] == r1["score"]
    assert np.array_equal(r0["nodes"], r1["nodes"]) and r0["npass"] == r1["npass"]
    # and it is the single-process answer
    p = synth.make_euclidean_problem(m, 0.9, seed=21)
    c = ref.RefClipper()
    c.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = c.solve(p.u0)
    assert r0["nodes"].tolist() == s.nodes.tolist()
    assert abs(float(r0["score"]) - s.score) <= 1e-9 * s.score


def test_shard_plan():
    from clipper_amd import dist

    for m, world in [(10000, 1), (10000, 8), (300000, 8), (1000, 3), (65, 4), (12, 8)]:
        W = dist.shard_pitch(m, world)
        assert W % 64 == 0 and W * world >= m
        cols = [dist.shard_columns(m, world, r) for r in range(world)]
        assert cols[0][0] == 0 and cols[-1][1] == m
        assert all(a1 == b0 for (_, a1), (b0, _) in zip(cols, cols[1:]))   # contiguous cover
        assert all(0 <= c1 - c0 <= W for c0, c1 in cols)
    # gathered layout round trip
    m, world = 1000, 3
    W = dist.shard_pitch(m, world)
    a, b = np.arange(m, dtype=float), -np.arange(m, dtype=float)
    blocks = []
    for r in range(world):
        c0, c1 = dist.shard_columns(m, world, r)
        blocks.append(dist.pack_block(a[c0:c1], b[c0:c1], W))
    a2, b2 = dist.unpack_gathered(np.concatenate(blocks), m, world)
    assert np.array_equal(a, a2) and np.array_equal(b, b2)
