#!/usr/bin/env python3
"""Wall-clock of clipper_hip_knn (host buffers in / out) at a few sizes; run under
rocprofv3 --kernel-trace --stats for the kernel times.  python tools/knn_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clipper_amd import _abi as abi

rng = np.random.default_rng(0)
for n, k in [(4096, 1), (10000, 1), (10000, 8), (100000, 1), (100000, 16)]:
    P0, P1 = rng.random((3, n)), rng.random((3, n))
    abi.knn(P0, P1, k)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); abi.knn(P0, P1, k); ts.append((time.perf_counter() - t0) * 1e3)
    print(f"n0=n1={n} knn={k}: {min(ts):.2f} ms per call (host buffers in/out), {n * n / min(ts) * 1e-6:.1f} G pairs/s")
