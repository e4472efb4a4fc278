#!/usr/bin/env python3
"""How long the exact densest-subgraph rounding (Rounding::DSD, clipper.cpp:294-300 -> dsd.cpp) takes at
the sizes the sweep produces: nnz(u) of the m = 10k / 30k / 100k solutions. The sub-matrix induced
by nnz(u) is gathered from the device (k_gather_sub), Goldberg's bisection + max-flow runs on the host.
  python tools/dsd_timing.py [--sizes 10000,30000,100000]"""
import argparse
import json
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
    ap.add_argument("--sizes", default="10000,30000,100000")
    a = ap.parse_args()
    for m in [int(x) for x in a.sizes.split(",")]:
        p = synth.make_euclidean_problem(m, 0.95)
        g = abi.HipClipper(storage=abi.STORE_F32_CSC)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        s = g.solve(p.u0)                      # DSD_HEU (default)
        S = np.flatnonzero(s.u > 0).astype(np.int32)
        t0 = time.perf_counter()
        nodes = g.densest_subgraph(S)
        t1 = time.perf_counter()
        g.params.rounding = abi.ROUNDING_DSD
        t2 = time.perf_counter()
        s2 = g.solve(p.u0)
        t3 = time.perf_counter()
        print(json.dumps(dict(m=m, heu_nodes=len(s.nodes), nnz_u=int(S.size), dsd_nodes=int(len(nodes)),
                              dsd_call_ms=round((t1 - t0) * 1e3, 2), solve_with_dsd_ms=round((t3 - t2) * 1e3, 2),
                              solve_heu_ms=round(s.t * 1e3, 2),
                              same_set=sorted(nodes.tolist()) == sorted(s.nodes.tolist()))), flush=True)
        g.close()


if __name__ == "__main__":
    main()
