import sys
sys.path.insert(0, '.')
import numpy as np
from clipper_amd import _abi as abi, synth
m, rho, seed = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
p = synth.make_euclidean_problem(m, rho, seed=seed)
for storage in (abi.STORE_F32_CSC, abi.STORE_F64_CSC):
    for mode in (1, 2, 0):
        g = abi.HipClipper(storage=storage)
        g.set_row_view(mode)
        g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        s = g.solve(p.u0)
        st = g.view_stats()
        print(f"storage {storage} mode {mode} (1 = no views, 2 = streamed views, 0 = default): passes {s.n_passes} trials {s.n_trials} ifinal {s.ifinal} "
              f"score {s.score!r} nodes {len(s.nodes)} | views {st.builds} rows {st.rows} view passes {st.view_passes} resident {st.resident_launches}", flush=True)
        g.close()
