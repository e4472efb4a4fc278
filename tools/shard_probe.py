#!/usr/bin/env python3
"""Per-shard cost of the column-sharded iteration on ONE GPU: P logical shards driven by one thread
(clipper_hip_create_group with a repeated device). Prints the solve time and the mean duration of
shard 0's pass kernel (decision + pass over its W columns).
  python tools/shard_probe.py [m] [P ...]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from clipper_amd import _abi as abi, synth

m = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
st_name = os.environ.get("PROBE_STORAGE", "f32")
ST = {"f32": abi.STORE_F32, "csc": abi.STORE_F32_CSC}[st_name]
Ps = [int(x) for x in sys.argv[2:]] or [1, 2, 4, 8]
p = synth.make_euclidean_problem(m, 0.95 if m > 1000 else 0.9)
for P in Ps:
    g = abi.HipClipper(storage=ST, group=[0] * P) if P > 1 else abi.HipClipper(storage=ST)
    g.stage_inputs(p.D1, p.D2, p.A)
    g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
    g.stage_u0(p.u0)
    g.solve_staged()
    g.set_profiling(True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); s = g.solve_staged(); ts.append((time.perf_counter() - t0) * 1e3)
    tm = g.timings()
    print(f"{st_name} m={m} P={P}: solve {min(ts):.3f} ms, passes {s.n_passes}, shard-0 pass kernel {tm.gemv_avg_us:.1f} us "
          f"(min {tm.gemv_min_us:.1f}), per pass wall {1e3*min(ts)/s.n_passes:.1f} us, nodes {len(s.nodes)}")
    g.close()
