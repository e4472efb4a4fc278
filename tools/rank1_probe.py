#!/usr/bin/env python3
"""The multi-process driver on ONE GPU (a 1-rank RCCL world, CLIPPER_HIP_FORCE_RCCL=1): what a
rank of `bench.py --gpus N` runs per solve except that the all-gather has nobody to talk to.
  CLIPPER_HIP_FORCE_RCCL=1 [CLIPPER_HIP_SOLVE_BATCH=n] python tools/rank1_probe.py [m] [storage]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipper_amd import _abi as abi, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
st = {"f32": abi.STORE_F32, "csc": abi.STORE_F32_CSC}[sys.argv[2] if len(sys.argv) > 2 else "csc"]
p = synth.make_euclidean_problem(m, 0.95 if m > 1000 else 0.9)
g = abi.HipClipper(storage=st, rank=0, world=1)
g.comm_init(g.unique_id())
g.stage_inputs(p.D1, p.D2, p.A)
g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
g.stage_u0(p.u0)
g.solve_staged()
ts = []
for _ in range(5):
    t0 = time.perf_counter(); s = g.solve_staged(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"m={m} storage={g.storage_in_use} batch={os.environ.get('CLIPPER_HIP_SOLVE_BATCH','default')}: "
      f"solve {min(ts):.3f} ms (median {sorted(ts)[2]:.3f}), passes {s.n_passes}, nodes {len(s.nodes)}")
