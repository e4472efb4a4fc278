#!/usr/bin/env python3
"""Where a window iteration of the resident solver goes (unit 0): wall-clock stamps inside the kernel.
  CLIPPER_HIP_STAMPS=1 python tools/resident_timeline.py [--sizes 100,1000,2048]
columns: candidates+norms | X table+wave pass | reduce+publish | poll | gather | evaluate+reduce  (us, medians)
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from clipper_amd import _abi as abi  # noqa: E402
from clipper_amd import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="100,1000,2048")
    a = ap.parse_args()
    for m in [int(x) for x in a.sizes.split(",")]:
        p = synth.make_euclidean_problem(m, 0.9)
        g = abi.HipClipper(storage=abi.STORE_F32_CSC)
        g.stage_inputs(p.D1, p.D2, p.A)
        g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
        g.stage_u0(p.u0)
        g.solve_staged()
        sol = g.solve_staged()
        st = g.debug_stamps().reshape(-1, 8)[:500]
        rows = st[(st[:, 0] > 0) & (st[:, 6] > 0)]
        d = np.diff(rows[:, :7], axis=1) / 100.0  # us
        med = np.median(d, axis=0)
        print(f"m={m} solver={g.last_solver} passes={sol.n_passes} windows={len(rows)} "
              f"cand+norms {med[0]:.2f} | table+pass {med[1]:.2f} | reduce+publish {med[2]:.2f} | poll {med[3]:.2f} "
              f"| gather {med[4]:.2f} | eval+reduce {med[5]:.2f} | window total {np.median(rows[:, 6] - rows[:, 0]) / 100:.2f} us")
        fx = g.debug_stamps().reshape(-1, 8)[500]
        print(f"   kernel: setup {(fx[1]-fx[0])/100:.2f} | init {(fx[2]-fx[1])/100:.2f} | iterations {(fx[3]-fx[2])/100:.2f} | hand-over {(fx[4]-fx[3])/100:.2f} us")
        g.close()


if __name__ == "__main__":
    main()
