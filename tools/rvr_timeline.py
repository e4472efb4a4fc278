#!/usr/bin/env python3
"""Where a turn of the resident solver on a row view goes (unit 0): wall-clock stamps inside the kernel
(csrc/k_rv_resident.hip.h), on the headline problem.
  CLIPPER_HIP_STAMPS=1 python tools/rvr_timeline.py [--m 10000] [--rho 0.95]
columns (us, medians over the turns): candidates + pass | tail + publish | exchange | scalars | decide
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from clipper_amd import _abi as abi  # noqa: E402
from clipper_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--m", type=int, default=10000)
    ap.add_argument("--rho", type=float, default=0.95)
    ap.add_argument("--storage", default="csc")
    ap.add_argument("--units", action="store_true", help="print every unit's pass duration of turn 6")
    a = ap.parse_args()
    p = synth.make_euclidean_problem(a.m, a.rho, seed=12345)
    g = abi.HipClipper(storage=abi.STORE_F32_CSC if a.storage == "csc" else abi.STORE_F64_CSC)
    g.stage_inputs(p.D1, p.D2, p.A)
    g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
    g.stage_u0(p.u0)
    for _ in range(3):
        g.solve_staged()
    ts = []
    for _ in range(7):
        t0 = time.perf_counter()
        sol = g.solve_staged()
        ts.append((time.perf_counter() - t0) * 1e3)
    st = g.view_stats()
    raw = g.debug_stamps().reshape(-1)
    rows = raw[:501 * 8].reshape(501, 8)
    end = rows[500]
    n = int(end[3])
    t = rows[:n, :5].astype(np.float64)
    d = np.diff(t, axis=1) / 100.0
    turn = np.diff(t[:, 0]) / 100.0
    if n < 2:
        print(f"m={a.m}: solve {np.median(ts):.3f} ms (min {min(ts):.3f}); passes {sol.n_passes}, no resident launch")
        g.close()
        return
    med = np.median(d, axis=0)
    print(f"m={a.m} wgs={os.environ.get('CLIPPER_HIP_VIEW_RESIDENT_WGS', 'auto')}: solve {np.median(ts):.3f} ms (min {min(ts):.3f}); "
          f"passes {sol.n_passes} ({st.view_passes} on a view of {st.rows} rows), resident launches {st.resident_launches}, turns {n}")
    print(f"   per turn: candidates+pass {med[0]:.2f} | tail+publish {med[1]:.2f} | exchange {med[2]:.2f} | scalars {med[3]:.2f} "
          f"| turn to turn {np.median(turn):.2f} (p90 {np.percentile(turn, 90):.2f}) us")
    per = raw[4096:4096 + 4 * 256].reshape(256, 4).astype(np.float64)
    per = per[per[:, 0] > 0]
    if len(per) and n > 6:
        t0 = per[:, 0].min()
        rel = (per - t0) / 100.0
        for name, c in (("turn start", 0), ("pass done", 1), ("published", 2), ("gathered", 3)):
            v = rel[:, c]
            print(f"   turn 6, all {len(per)} units, {name:11s}: min {v.min():6.2f}  p50 {np.median(v):6.2f}  p90 {np.percentile(v, 90):6.2f}  max {v.max():6.2f} us"
                  + (f"   (slowest units: {np.argsort(-v)[:6].tolist()})" if c == 1 else ""))
        pd = rel[:, 1] - rel[:, 0]
        if a.units:   # every unit's pass of turn 6 (beside CLIPPER_HIP_RESIDENT_DEBUG=2's plan lines on stderr)
            for i, v in enumerate(pd):
                print(f"unit {i} pass_us {v:.2f}")
        print(f"   turn 6, pass duration per unit: min {pd.min():.2f} p50 {np.median(pd):.2f} p90 {np.percentile(pd, 90):.2f} max {pd.max():.2f} us")
    print(f"   launch: slices -> LDS {(end[1] - end[0]) / 100:.2f} us, loop {(end[2] - end[1]) / 100:.2f} us, total {(end[2] - end[0]) / 100:.2f} us")
    g.close()


if __name__ == "__main__":
    main()
