#!/usr/bin/env python3
"""Row view on / off at several sizes (one GPU): solve wall clock, passes, how many of them streamed a
view, the views built, bit-reproducibility of u across repeated solves. No oracle leg.
  python tools/rowview_probe.py --m 10000 30000 100000 [--storage csc|csc64] [--reps 3]"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from clipper_amd import _abi as abi  # noqa: E402
from clipper_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, nargs="+", default=[10000, 30000])
    ap.add_argument("--rho", type=float, default=0.95)
    ap.add_argument("--storage", default="csc")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    a = ap.parse_args()
    storage = {"csc": abi.STORE_F32_CSC, "csc64": abi.STORE_F64_CSC}[a.storage]
    for m in a.m:
        p = synth.make_euclidean_problem(m, a.rho, seed=12345)
        out = {"m": m, "rho": a.rho, "storage": a.storage}
        for mode, name in ((1, "off"), (0, "on")):
            g = abi.HipClipper(storage=storage)
            g.set_row_view(mode)
            if a.profile:
                g.set_profiling(True)
            g.stage_inputs(p.D1, p.D2, p.A)
            g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
            aff_ms = g.timings().affinity_kernel_ms
            g.stage_u0(p.u0)
            times, hashes = [], set()
            for _ in range(a.reps + 1):
                t0 = time.perf_counter()
                s = g.solve_staged()
                times.append((time.perf_counter() - t0) * 1e3)
                hashes.add(hashlib.sha256(np.ascontiguousarray(s.u).tobytes()).hexdigest()[:16])
            st = g.view_stats()
            tm = g.timings()
            out[name] = dict(solve_ms=round(float(np.median(times[1:])), 4), first_ms=round(times[0], 4),
                             affinity_kernel_ms=round(aff_ms, 4), passes=int(s.n_passes), trials=int(s.n_trials),
                             ifinal=int(s.ifinal), score=float(s.score), nodes=int(len(s.nodes)),
                             nodes_sha=hashlib.sha256(np.sort(np.asarray(s.nodes, np.int32)).tobytes()).hexdigest()[:16],
                             u_hashes=sorted(hashes), builds=int(st.builds), rows=int(st.rows), view_bytes=int(st.bytes),
                             view_passes=int(st.view_passes), build_ms=round(st.build_ms, 4),
                             pass_us=round(tm.gemv_avg_us, 2), view_pass_us=round(st.view_pass_avg_us, 2),
                             slice_bytes=float(tm.gemv_bytes))
            g.close()
        out["speedup"] = round(out["off"]["solve_ms"] / out["on"]["solve_ms"], 3)
        out["same_nodes"] = out["off"]["nodes_sha"] == out["on"]["nodes_sha"]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
