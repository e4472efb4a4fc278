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
allst = g.debug_stamps()
fill = allst.reshape(-1)[8192:8192 + 4000].reshape(800, 5)
st = allst[:1536]
st = st[st[:, 2] > 0]
t0 = st[:, 0].min()
pc = lambda v, f: float(np.sort(v)[int(f * (len(v) - 1))])
start, head, end = (st[:, 0] - t0) * 0.01, (st[:, 1] - st[:, 0]) * 0.01, (st[:, 2] - t0) * 0.01
body = (st[:, 2] - st[:, 1]) * 0.01
vs = g.view_stats()
print(f"m={m} workgroups stamped {len(st)} passes {s.n_passes} (on a row view: {vs.view_passes}; last view {vs.rows} rows, {vs.bytes} bytes)")
for name, v in (("start", start), ("head (launch -> decision done)", head), ("body", body), ("end", end)):
    print(f"  {name:32s} p10 {pc(v,.1):6.2f} p50 {pc(v,.5):6.2f} p90 {pc(v,.9):6.2f} p99 {pc(v,.99):6.2f} max {v.max():6.2f} us")
work = (st[:, 3] >> 32) > 0   # workgroups that had chunks to stream (a pass on a view leaves most of the grid idle)
if work.sum() and work.sum() < len(st):
    print(f"  workgroups with chunks to stream: {int(work.sum())}")
    for name, v in (("head", head[work]), ("body", body[work]), ("end", end[work])):
        print(f"    {name:30s} p10 {pc(v,.1):6.2f} p50 {pc(v,.5):6.2f} p90 {pc(v,.9):6.2f} p99 {pc(v,.99):6.2f} max {v.max():6.2f} us")
tl = allst[1536:2048]
tl = tl[tl[:, 2] > 0]
if len(tl):
    t1 = tl[:, 0].min()
    print(f"tail workgroups stamped {len(tl)}; tail starts {(t1 - t0) * 0.01:.2f} us after the pass started, "
          f"{(t1 - st[:, 2].max()) * 0.01:.2f} us after its last workgroup ended")
    for name, v in (("start", (tl[:, 0] - t1) * 0.01), ("state + slot sums", (tl[:, 1] - tl[:, 0]) * 0.01),
                    ("elementwise + reduce", (tl[:, 2] - tl[:, 1]) * 0.01), ("end", (tl[:, 2] - t1) * 0.01)):
        print(f"  {name:32s} p10 {pc(v,.1):6.2f} p50 {pc(v,.5):6.2f} p90 {pc(v,.9):6.2f} p99 {pc(v,.99):6.2f} max {v.max():6.2f} us")
fill = fill[fill[:, 4] > 0]
if len(fill):
    f0 = fill[:, 0].min()
    print(f"fill kernel, first {len(fill)} tiles (of the first round of workgroups):")
    for name, v in (("start", (fill[:, 0] - f0) * 0.01), ("prefilter + exact scores", (fill[:, 1] - fill[:, 0]) * 0.01),
                    ("barrier + masks + sizes", (fill[:, 2] - fill[:, 1]) * 0.01), ("claim (atomic) + barrier", (fill[:, 3] - fill[:, 2]) * 0.01),
                    ("write the steps", (fill[:, 4] - fill[:, 3]) * 0.01), ("tile total", (fill[:, 4] - fill[:, 0]) * 0.01)):
        print(f"  {name:32s} p10 {pc(v,.1):6.2f} p50 {pc(v,.5):6.2f} p90 {pc(v,.9):6.2f} p99 {pc(v,.99):6.2f} max {v.max():6.2f} us")
