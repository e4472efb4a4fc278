#!/usr/bin/env python3
"""First thing to run on a multi-GPU box: does the column-sharded path work here, step by step?

  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port 29511 \
         tools/multigpu_preflight.py [--size 20000] [--exchange rccl|callback] [--same-device]

One rank per GPU (`--same-device`: every rank on device 0 — with `--exchange callback` that is how the
multi-process driver runs on a ONE-GPU box). Every step prints PASS / FAIL with what it saw; the
first FAIL names the layer that is broken:
  1 environment      HSA_ENABLE_IPC_MODE_LEGACY, visible devices, librccl
  2 rendezvous       torch.distributed (gloo) barrier + broadcast of the RCCL unique id
  3 communicator     clipper_hip_create_rank + clipper_hip_comm_init (ncclCommInitRank) or the callback
  4 exchange         one all-gather of a [2][W] block through the library's own exchange (the matvec
                     API: per-rank fill of M's column shard, one pair-mode pass, reduce, all-gather):
                     identical gathered products on every rank, equal to a single-GPU product on rank 0
  5 solve            per-rank fill + the whole solver (one [V+1][W] all-gather per pass): identical u
                     hashes on every rank, the single-GPU node list on every rank; pass / exchange times
  6 row views        the same solve built a row view on EVERY rank (each its own columns of the same rows) and
                     ran passes on it: the same build and view-pass counts everywhere (m >= 3000)
"""
import argparse
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", dest="m", type=int, default=20000, help="associations (not --m: torch.distributed.run would read it as one of its own options)")
    ap.add_argument("--exchange", choices=["rccl", "callback"], default="rccl")
    ap.add_argument("--same-device", action="store_true")
    ap.add_argument("--storage", choices=["csc", "csc64", "f32"], default="csc")
    a = ap.parse_args()
    rank, local_rank, world = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("LOCAL_RANK", "0"), ("WORLD_SIZE", "1")))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29511")
    ok_all = True

    def report(step, ok, msg):
        nonlocal ok_all
        ok_all = ok_all and ok
        print(f"[rank {rank}] step {step}: {'PASS' if ok else 'FAIL'} — {msg}", flush=True)

    # 1 environment
    import torch
    import torch.distributed as tdist
    ndev = torch.cuda.device_count()
    ipc = os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    dev = 0 if a.same_device else local_rank
    report(1, ndev > dev and (ipc == "0" or a.exchange == "callback"),
           f"{ndev} device(s) visible, this rank uses {dev}; HSA_ENABLE_IPC_MODE_LEGACY={ipc!r} "
           f"(must be '0' for RCCL between processes: the host driver only supports dmabuf IPC)")
    # 2 rendezvous
    t0 = time.time()
    tdist.init_process_group("gloo", rank=rank, world_size=world)
    tdist.barrier()
    report(2, True, f"gloo rendezvous of {world} rank(s) in {time.time() - t0:.2f} s")
    from clipper_amd import _abi as abi
    from clipper_amd import dist as cdist
    from clipper_amd import synth
    storage = {"csc": abi.STORE_F32_CSC, "csc64": abi.STORE_F64_CSC, "f32": abi.STORE_F32}[a.storage]
    # 3 communicator
    try:
        if world == 1:
            os.environ["CLIPPER_HIP_FORCE_RCCL"] = "1"   # a 1-rank world still goes through ncclAllGather
        g = abi.HipClipper(device=dev, storage=storage, rank=rank, world=world)
        if a.exchange == "rccl":
            uid = cdist.broadcast_bytes(g.unique_id() if rank == 0 else None, 128, src=0)
            g.comm_init(uid)
            what = "ncclCommInitRank inside libclipper_hip.so"
        else:
            def allgather(block):
                t = torch.from_numpy(block)
                out = [torch.empty_like(t) for _ in range(world)]
                tdist.all_gather(out, t)
                return np.concatenate([o.numpy() for o in out])
            g.comm_init_callback(allgather)
            what = "exchange through a gloo all-gather callback"
        report(3, True, what)
    except Exception as e:   # noqa: BLE001
        report(3, False, f"{type(e).__name__}: {e}")
        raise SystemExit(1)
    # 4 exchange: one product
    p = synth.make_euclidean_problem(a.m, 0.95, seed=4711)
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    x = np.random.default_rng(5).random(a.m)
    yM, yC = g.matvec(x)
    h = hashlib.sha256(yM.tobytes() + yC.tobytes()).hexdigest()[:16]
    hs = [None] * world
    tdist.all_gather_object(hs, h)
    same = len(set(hs)) == 1
    ref_ok, ref_msg = True, ""
    if rank == 0:
        g1 = abi.HipClipper(device=dev, storage=storage)
        g1.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        zM, zC = g1.matvec(x)
        err = float(np.max(np.abs(yM - zM)) / max(1.0, np.max(np.abs(zM))))
        ref_ok = err <= 1e-12 and np.max(np.abs(yC - zC)) <= 1e-9
        ref_msg = f"; vs one GPU: rel {err:.1e}"
    report(4, same and ref_ok, f"gathered products {'identical' if same else 'DIFFER'} across ranks {sorted(set(hs))}{ref_msg}")
    # 5 solve
    g.set_profiling(True)
    t0 = time.time()
    s = g.solve(p.u0)
    dt = time.time() - t0
    tm = g.timings()
    hu = hashlib.sha256(np.ascontiguousarray(s.u).tobytes()).hexdigest()[:16]
    hus = [None] * world
    tdist.all_gather_object(hus, hu)
    same = len(set(hus)) == 1
    ref_ok, ref_msg = True, ""
    if rank == 0:
        s1 = g1.solve(p.u0)
        ref_ok = sorted(s1.nodes.tolist()) == sorted(s.nodes.tolist()) and abs(s1.score - s.score) <= 1e-8 * abs(s1.score)
        ref_msg = (f"; one GPU: {len(s1.nodes)} nodes, score {s1.score:.9f} — this world: {len(s.nodes)} nodes, "
                   f"score {s.score:.9f}")
        g1.close()
    report(5, same and ref_ok,
           f"u {'identical' if same else 'DIFFERS'} across ranks; {s.n_passes} passes in {dt * 1e3:.1f} ms, pass on this rank's "
           f"shard {tm.gemv_avg_us:.1f} us ({tm.gemv_bytes / 1e6:.1f} MB), exchange {tm.exchange_avg_us:.1f} us "
           f"({tm.exchange_bytes / 1e3:.1f} KB per rank, {tm.exchange_samples} sampled){ref_msg}")
    # 6 row views: every rank built its columns of the same views and ran the same passes on them
    st = g.view_stats()
    vs = [None] * world
    tdist.all_gather_object(vs, (int(st.builds), int(st.rows), int(st.view_passes), int(st.resident_launches), int(st.resident_giveups)))
    want_views = a.m >= 3000 and a.storage != "f32"
    report(6, (len(set(vs)) == 1) and (not want_views or (st.builds >= 1 and st.view_passes >= 1)),
           f"views built {st.builds} (last: {st.rows} rows, {st.bytes / 1e6:.1f} MB on this rank), passes on a view {st.view_passes} of "
           f"{st.passes}, view pass on this rank's shard {st.view_pass_avg_us:.1f} us; resident launches on the view's replica "
           f"{st.resident_launches} (gave up: {st.resident_giveups}); per rank (builds, rows, view passes, resident launches, give-ups): {vs}")
    g.close()
    tdist.barrier()
    tdist.destroy_process_group()
    if rank == 0:
        print("PREFLIGHT " + ("OK" if ok_all else "FAILED"), flush=True)
    raise SystemExit(0 if ok_all else 1)


if __name__ == "__main__":
    main()
