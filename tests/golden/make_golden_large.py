#!/usr/bin/env python3
"""Golden answers of the CPU oracle (oracle/clipper_ref.cpp) at sizes the test suite cannot
afford to recompute: the north-star size m = 100 000 (and m = 30 000 as a cross-check that the
suite also recomputes). Run on any host with ~16 GB of RAM:

    python tests/golden/make_golden_large.py 100000 > tests/golden/oracle_m100000.json

Writes the problem's defining parameters (the inputs are regenerated from them by
clipper_amd.synth, bit for bit), the selected node list AS PRODUCED (ordered, with u at every listed
node: entries of u that differ in the last bits may swap places under another order of the partial
sums — the test compares the lists up to such swaps), its SHA-256 and length, the
objective, the iteration counters and the stored-entry count. tests/test_gpu_configs.py asserts the GPU path
against it. The oracle, not the reference binary (unbuildable here: no Eigen) — see DESIGN.md."""
import hashlib
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np

from clipper_amd import synth
from oracle import clipper_ref as ref


def main():
    m = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
    rho, seed = 0.95, 12345
    p = synth.make_euclidean_problem(m, rho, seed=seed)
    r = ref.RefClipper()
    t0 = time.time()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    t1 = time.time()
    s = r.solve(p.u0)
    t2 = time.time()
    nodes = np.asarray(s.nodes, dtype=np.int32)
    out = dict(m=m, rho=rho, seed=seed, invariant=dict(kind="EuclideanDistance", **synth.EUCLID_BENCH_PARAMS),
               params="clipper::Params defaults", u0="default_rng(seed + 1).random(m)",
               nnz_upper=int(r.nnz), num_nodes=int(nodes.size),
               nodes_sha256=hashlib.sha256(nodes.tobytes()).hexdigest(),
               nodes_sorted_sha256=hashlib.sha256(np.sort(nodes).tobytes()).hexdigest(),
               first_nodes=nodes[:8].tolist(), nodes=nodes.tolist(),
               u_at_nodes=[float(x).hex() for x in np.asarray(s.u, dtype=np.float64)[nodes]],
               nnz_u=int(np.count_nonzero(np.asarray(s.u) > 0)),
               score=float(s.score), ifinal=int(s.ifinal),
               n_trials=int(getattr(s, "n_trials", -1)), n_passes=int(getattr(s, "n_passes", -1)),
               u_sha256=hashlib.sha256(np.asarray(s.u, dtype=np.float64).tobytes()).hexdigest(),
               oracle_affinity_s=round(t1 - t0, 2), oracle_solve_s=round(t2 - t1, 2),
               oracle_threads=ref.omp_threads(), host=os.uname().nodename,
               generated_by="tests/golden/make_golden_large.py")
    json.dump(out, sys.stdout, indent=None, separators=(",", ":"))
    print()


if __name__ == "__main__":
    main()
