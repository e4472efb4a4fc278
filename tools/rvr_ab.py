import sys, time
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth
for (m, rho, seed) in [(12544, 0.95, 805946), (14257, 0.97, 3902), (10000, 0.95, 12345), (16000, 0.96, 7), (20000, 0.97, 8)]:
    p = synth.make_euclidean_problem(m, rho, seed=seed)
    row = []
    for mode in (0, 2):
        g = abi.HipClipper(storage=abi.STORE_F32_CSC)
        g.set_row_view(mode)
        g.stage_inputs(p.D1, p.D2, p.A)
        g.affinity_euclidean_staged(**synth.EUCLID_BENCH_PARAMS)
        g.stage_u0(p.u0)
        for _ in range(3): g.solve_staged()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); s = g.solve_staged(); ts.append((time.perf_counter() - t0) * 1e3)
        st = g.view_stats()
        row.append((np.median(ts), st.rows, st.resident_launches, s.n_passes, st.view_passes, s.n_trials))
        g.close()
    print(m, rho, 'resident-capable:', row[0], '| streamed:', row[1])
