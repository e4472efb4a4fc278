#!/usr/bin/env python3
"""Where the time of one pass launch goes: per-workgroup {start, decision done, end} stamps of the
last pass launch of a solve at the headline problem (CLIPPER_HIP_STAMPS=1). Measurement only."""
import os
import sys

os.environ["CLIPPER_HIP_STAMPS"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from clipper_amd import _abi as abi
from clipper_amd import synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
p = synth.make_euclidean_problem(m, 0.95 if m >= 10000 else 0.9, seed=12345)
g = abi.HipClipper(device=0, storage=abi.STORE_F32_CSC)
for rep in range(3):
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = g.solve(p.u0)
st = g.debug_stamps()
st = st[st[:, 2] > 0]
t0 = st[:, 0].min()
pc = lambda v, f: float(np.sort(v)[int(f * (len(v) - 1))])
start, head, end = (st[:, 0] - t0) * 0.01, (st[:, 1] - st[:, 0]) * 0.01, (st[:, 2] - t0) * 0.01
body = (st[:, 2] - st[:, 1]) * 0.01
print(f"m={m} workgroups stamped {len(st)} passes {s.n_passes}")
for name, v in (("start", start), ("head (launch -> decision done)", head), ("body", body), ("end", end)):
    print(f"  {name:32s} p10 {pc(v,.1):6.2f} p50 {pc(v,.5):6.2f} p90 {pc(v,.9):6.2f} p99 {pc(v,.99):6.2f} max {v.max():6.2f} us")
