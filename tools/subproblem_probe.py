#!/usr/bin/env python3
"""The live sub-problem on / off at several sizes (one GPU): solve wall clock, passes, how many of them ran on a row
view / on the sub-problem, bit-reproducibility of u across repeated solves, agreement of the three routes (no views,
views, views + sub-problem). No oracle leg.
  python tools/subproblem_probe.py --m 30000 100000 [--storage csc|csc64] [--reps 3] [--profile]"""
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
    ap.add_argument("--m", type=int, nargs="+", default=[30000])
    ap.add_argument("--rho", type=float, default=0.95)
    ap.add_argument("--storage", default="csc")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--seed", type=int, default=12345)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--modes", default="views,sub")
    a = ap.parse_args()
    storage = {"csc": abi.STORE_F32_CSC, "csc64": abi.STORE_F64_CSC}[a.storage]
    for m in a.m:
        p = synth.make_euclidean_problem(m, a.rho, seed=a.seed)
        out = {"m": m, "rho": a.rho, "storage": a.storage}
        for name in a.modes.split(","):
            g = abi.HipClipper(storage=storage)
            # noviews | views (no sub-problem) | sub (the default route) | sub_streamed (views never resident: the
            # sub-problem also where the resident launch would have taken the view)
            g.set_row_view(1 if name == "noviews" else (2 if name == "sub_streamed" else 0))
            g.set_subproblem(0 if name in ("sub", "sub_streamed") else (2 if name == "sub_slices" else 1))
            if a.profile:
                g.set_profiling(True)
            g.stage_inputs(p.D1, p.D2, p.A)
            g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
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
                             passes=int(s.n_passes), trials=int(s.n_trials), ifinal=int(s.ifinal), score=float(s.score),
                             nodes=int(len(s.nodes)),
                             nodes_sha=hashlib.sha256(np.sort(np.asarray(s.nodes, np.int32)).tobytes()).hexdigest()[:16],
                             order_sha=hashlib.sha256(np.asarray(s.nodes, np.int32).tobytes()).hexdigest()[:16],
                             u_hashes=sorted(hashes), builds=int(st.builds), rows=int(st.rows), view_passes=int(st.view_passes),
                             view_build_ms=round(st.build_ms, 3), resident_launches=int(st.resident_launches), pass_us=round(tm.gemv_avg_us, 2),
                             view_pass_us=round(st.view_pass_avg_us, 2), sub_entries=int(st.sub_entries),
                             sub_leaves=int(st.sub_leaves), sub_passes=int(st.sub_passes), sub_rows=int(st.sub_rows),
                             sub_bytes=int(st.sub_bytes), sub_build_ms=round(st.sub_build_ms, 3),
                             sub_pass_us=round(st.sub_pass_avg_us, 2), sub_pass_samples=int(st.sub_pass_samples),
                             sub_dense=int(st.sub_dense))
            out[name + "_u"] = s.u
            g.close()
        names = [n for n in a.modes.split(",")]
        if len(names) > 1:
            ref = names[0]
            for n in names[1:]:
                # (same_order: the selected LIST as produced — near-equal entries of u trade places when partial sums
                # associate differently; the suite compares lists up to such ties)
                out[f"{n}_vs_{ref}"] = dict(same_nodes=out[n]["nodes_sha"] == out[ref]["nodes_sha"],
                                            same_order=out[n]["order_sha"] == out[ref]["order_sha"],
                                            same_ifinal=out[n]["ifinal"] == out[ref]["ifinal"],
                                            rel_dscore=abs(out[n]["score"] - out[ref]["score"]) / abs(out[ref]["score"]),
                                            max_du=float(np.max(np.abs(out[n + "_u"] - out[ref + "_u"]))),
                                            speedup=round(out[ref]["solve_ms"] / out[n]["solve_ms"], 3))
        for n in names:
            del out[n + "_u"]
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
