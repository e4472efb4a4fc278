#!/bin/bash
# feasibility of the column-compressed pass on the real affinity pattern. usage: tools/gpu_csc.sh <tag> [m]
TAG=$1; M=${2:-10000}
OUT=gpurun_out/$TAG; mkdir -p $OUT
python - > $OUT/dump.log 2>&1 <<PY
import numpy as np, torch
from clipper_amd import _abi as abi, synth
abi.load_library()
p = synth.make_euclidean_problem($M, 0.95, seed=12345)
g = abi.HipClipper()
g.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
Mx = np.asarray(g.get_affinity_matrix(), dtype=np.float32)
np.fill_diagonal(Mx, 0)
print("density", (Mx != 0).mean())
cnt = (Mx != 0).sum(axis=0)
print("col nnz min/mean/max", cnt.min(), cnt.mean(), cnt.max())
Mx.tofile("/tmp/S.f32")
PY
cat $OUT/dump.log
for v in pf0 pf1; do timeout 300 tools/_bin/csc_tune_$v $M /tmp/S.f32 2>&1 | tee -a $OUT/real.log; done


