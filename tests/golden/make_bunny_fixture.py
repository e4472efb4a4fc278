#!/usr/bin/env python3
"""Extracts a small, deterministic sample of the Stanford-bunny model the reference ships
(/root/reference/examples/data/bun10k.ply, 9 992 float32 vertices) into
tests/golden/bunny_points.json — the fixture of BASELINE.json's cfg1 "bunny plumbing" case —
and a 4096-vertex sample (bunny_points_4096.f32) for the reference benchmark's table.
Run in the authoring container only (the reference tree does not exist on the GPU box):
    python tests/golden/make_bunny_fixture.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from clipper_amd import registration as reg  # noqa: E402

SRC = "/root/reference/examples/data/bun10k.ply"
N = 600

pts = reg.read_ply_xyz(SRC)
rng = np.random.default_rng(20240926)
idx = np.sort(rng.choice(len(pts), N, replace=False))
sample = pts[idx].astype(np.float32)
out = {
    "source": "examples/data/bun10k.ply of mit-acl/clipper (Stanford bunny, %d vertices)" % len(pts),
    "how": "read_ply_xyz + default_rng(20240926).choice(%d, %d, replace=False), sorted, float32" % (len(pts), N),
    "n_source_vertices": int(len(pts)),
    "bbox_min": pts.min(axis=0).round(6).tolist(),
    "bbox_max": pts.max(axis=0).round(6).tolist(),
    "points": [[float(v) for v in p] for p in sample],
}
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "bunny_points.json"), "w"))
print(len(pts), "vertices ->", N, "sampled; bbox", out["bbox_min"], out["bbox_max"])

# a larger sample for the reference benchmark's table (benchmarks/main.cpp: up to m = 2048
# associations at 0 % outliers need >= 2048 matched points): 4096 vertices, raw little-endian
# float32 x, y, z (48 KiB), same generator stream
N2 = 4096
idx2 = np.sort(np.random.default_rng(20240927).choice(len(pts), N2, replace=False))
pts[idx2].astype("<f4").tofile(os.path.join(ROOT, "tests", "golden", "bunny_points_4096.f32"))
print(N2, "vertices -> bunny_points_4096.f32")
