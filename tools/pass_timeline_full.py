#!/usr/bin/env python3
"""Every workgroup of the LAST pass launch of a solve: {start, decision done, end, chunks, where it ran}
(CLIPPER_HIP_STAMPS=2). What a launch's duration is made of: rounds, ramps, the spread of the items, the XCDs.
  python tools/pass_timeline_full.py [m=100000] [views=1]
Measurement only."""
import os
import sys

os.environ["CLIPPER_HIP_STAMPS"] = "2"
m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
views = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if not views:
    os.environ["CLIPPER_HIP_ROW_VIEW"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from clipper_amd import _abi as abi
from clipper_amd import synth

p = synth.make_euclidean_problem(m, 0.95 if m >= 10000 else 0.9, seed=12345)
g = abi.HipClipper(device=0, storage=abi.STORE_F32_CSC)
if views:
    g.set_row_view(2)   # streamed views only (the resident solver has its own stamps)
for rep in range(2):
    g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    s = g.solve(p.u0)
vs = g.view_stats()
st = g.debug_stamps()
st = st[st[:, 2] > 0]
t0 = st[:, 0].min()
start, head, end = (st[:, 0] - t0) * 0.01, (st[:, 1] - st[:, 0]) * 0.01, (st[:, 2] - t0) * 0.01
body = (st[:, 2] - st[:, 1]) * 0.01
chunks = (st[:, 3] >> 32).astype(np.int64)
xcd = ((st[:, 3] >> 8) & 0xF).astype(np.int64)
hwid = ((st[:, 3] >> 12) & 0xFFFF).astype(np.int64)
cu = xcd * 64 + ((hwid >> 13) & 7) * 16 + ((hwid >> 8) & 0xF)   # XCD, SE_ID, CU_ID (SH_ID folded into the CU bits' top)
pc = lambda v, f: float(np.sort(v)[int(f * (len(v) - 1))])
phase = (st[:, 3] & 0xFF).astype(np.int64)
print('phases of the stamped workgroups (3 = window pass):', dict(zip(*np.unique(phase, return_counts=True))))
print(f"m={m} views={'on' if views else 'off'}: {len(st)} workgroups stamped, {int((chunks > 0).sum())} with chunks; passes {s.n_passes} "
      f"({vs.view_passes} on a view of {vs.rows} rows, {vs.bytes} bytes); span of the launch {end.max():.1f} us")
w = chunks > 0
for name, v in (("start", start[w]), ("head", head[w]), ("body", body[w]), ("end", end[w]), ("chunks", chunks[w].astype(float)),
                ("body per chunk", body[w] / np.maximum(1, chunks[w]))):
    print(f"  {name:16s} p1 {pc(v,.01):8.2f} p10 {pc(v,.1):8.2f} p50 {pc(v,.5):8.2f} p90 {pc(v,.9):8.2f} p99 {pc(v,.99):8.2f} max {v.max():8.2f}")
# streaming workgroups and chunk throughput over time
edges = np.arange(0.0, end.max() + 5.0, 5.0)
print("  time (us)   workgroups in their body   in their head   chunks finished per us (by the items' mean rate)")
rate = chunks / np.maximum(body, 1e-9)
for a, b in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (a + b)
    s0 = (st[:, 1] - t0) * 0.01
    inb = (s0 <= mid) & (end > mid) & w
    inh = (start <= mid) & (s0 > mid)
    print(f"  {a:6.0f}-{b:<6.0f} {int(inb.sum()):8d} {int(inh.sum()):18d} {rate[inb].sum():22.1f}")
# rounds: the order in which the dispatcher started the workgroups
order = np.argsort(start)
first = order[: min(1536, len(order))]
late = order[min(1536, len(order)):]
print(f"  first 1536 started by {start[first].max():.2f} us; the other {len(late)} between {start[late].min() if len(late) else 0:.2f} and {start[late].max() if len(late) else 0:.2f} us")
print("  per XCD: workgroups, chunks, last end (us), mean body per chunk (us)")
for x in sorted(set(xcd.tolist())):
    k = (xcd == x) & w
    print(f"    xcd {x}: {int(k.sum()):5d} {int(chunks[k].sum()):7d} {end[k].max():8.1f} {float((body[k] / np.maximum(1, chunks[k])).mean()):8.2f}")
# per CU slot: how many workgroups each CU ran, how long it was busy
cus = sorted(set(cu.tolist()))
per = np.array([int(((cu == c) & w).sum()) for c in cus])
lastend = np.array([end[(cu == c)].max() for c in cus])
print(f"  {len(cus)} distinct (XCD, SE, CU) ids; workgroups with chunks per id: min {per.min()} p50 {int(np.median(per))} max {per.max()}; "
      f"last end per id: p10 {pc(lastend,.1):.1f} p50 {pc(lastend,.5):.1f} max {lastend.max():.1f} us")
# the ten workgroups that ended last
tail = np.argsort(-end)[:10]
print("  the ten that ended last: (block, start, head, body, chunks, xcd)")
for i in tail:
    print(f"    {int(i):5d} {start[i]:8.2f} {head[i]:6.2f} {body[i]:8.2f} {int(chunks[i]):4d} {int(xcd[i])}")
