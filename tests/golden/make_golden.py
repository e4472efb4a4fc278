#!/usr/bin/env python3
"""Extract the reference's own golden vectors for the hot path into
tests/golden/reference_vectors.json.

Run in the authoring container only (needs /root/reference, which does not exist on
the GPU box); the JSON it writes is committed and is what the tests read.

Sources (all literals are parsed out of the reference files, nothing is retyped):
  test/affinity_test.cpp:33-48   model points, rotation pi/8 about z, translation (5,3,0)
  test/affinity_test.cpp:93-106  exact 12x12 affinity matrix "from MATLAB"
  test/dsd_test.cpp:17-36        20x20 weighted affinity matrix (also sdp_test.cpp:17-36)
  test/dsd_test.cpp:15           DSD known answer {3,5,12,14,15} (kept for later rounds)
  examples/matlab/ex3_planecloud.m:18-33  plane parameters D1, D2; ground truth Agt
  examples/matlab/ex3_planecloud.m:79-86  PointNormalDistance params sign=deg2rad(1.5), epsn=1
"""
import json
import math
import os
import re

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_vectors.json")

NUM = r"[-+]?(?:\d+\.\d*|\.\d+|\d+)(?:[eE][-+]?\d+)?"


def numbers(text):
    return [float(x) for x in re.findall(NUM, text)]


def main():
    out = {}

    # ---- affinity_test.cpp ------------------------------------------------------------
    src = open(os.path.join(REF, "test/affinity_test.cpp")).read()
    cols = re.findall(r"model\.col\((\d)\)\s*<<\s*([^;]+);", src)
    model = np.zeros((3, 4))
    for c, vals in cols:
        model[:, int(c)] = numbers(vals)
    ang = re.search(r"AngleAxisd\(M_PI/(\d+),\s*Eigen::Vector3d::UnitZ\(\)\)", src)
    theta = math.pi / float(ang.group(1))
    t = np.array(numbers(re.search(r"translation\(\)\s*<<\s*([^;]+);", src).group(1)))
    R = np.array([[math.cos(theta), -math.sin(theta), 0.0],
                  [math.sin(theta), math.cos(theta), 0.0],
                  [0.0, 0.0, 1.0]])
    # data = T_MD.inverse() * model ; then conservativeResize(3,3) drops the last point
    data = (R.T @ (model - t[:, None]))[:, :3]
    mtrue_txt = re.search(r"Mtrue\s*<<\s*([^;]+);", src).group(1)
    Mtrue = np.array(numbers(mtrue_txt)).reshape(12, 12)
    assert np.array_equal(Mtrue, Mtrue.T) and np.all(np.diag(Mtrue) == 1)
    out["affinity_test"] = {
        "source": "test/affinity_test.cpp:33-48,93-106",
        "model": model.tolist(), "data": data.tolist(),
        "theta": theta, "translation": t.tolist(),
        "invariant": {"sigma": 0.01, "epsilon": 0.06, "mindist": 0.0},
        "Mtrue": Mtrue.tolist(),
        "expected_inlier_nodes": [0, 4, 8],  # clipper_test.cpp:62-66: A(i,0)==A(i,1), 3 rows
    }

    # ---- dsd_test.cpp 20x20 -------------------------------------------------------------
    src = open(os.path.join(REF, "test/dsd_test.cpp")).read()
    body = re.search(r"TEST\(DSD, Solve\)(.*?)std::vector<int> nodes", src, re.S).group(1)
    m20 = np.array(numbers(re.search(r"M\s*<<\s*([^;]+);", body).group(1))).reshape(20, 20)
    assert np.array_equal(m20, m20.T)
    dsd_nodes = [int(x) for x in numbers(re.search(r"dsd_nodes\s*=\s*\{([^}]+)\}", body).group(1))]
    out["dsd_test_20x20"] = {
        "source": "test/dsd_test.cpp:15-36 (same matrix in test/sdp_test.cpp:17-40, C=(M>0))",
        "M": m20.tolist(), "dsd_nodes": dsd_nodes,
    }

    # ---- ex3_planecloud.m ---------------------------------------------------------------
    src = open(os.path.join(REF, "examples/matlab/ex3_planecloud.m")).read()
    d1 = np.array(numbers(re.search(r"D1 = \[(.*?)\]'", src, re.S).group(1))).reshape(4, 4).T
    d2 = np.array(numbers(re.search(r"D2 = \[(.*?)\]'", src, re.S).group(1))).reshape(4, 4).T
    agt = np.array(numbers(re.search(r"Agt = \[([^\]]+)\]", src).group(1)), dtype=int).reshape(-1, 2)
    # DD = [zeros(3,n); D(1:3,:)]  (ex3_planecloud.m:84-85): zero points, plane normals
    DD1 = np.vstack([np.zeros((3, d1.shape[1])), d1[:3, :]])
    DD2 = np.vstack([np.zeros((3, d2.shape[1])), d2[:3, :]])
    out["planecloud"] = {
        "source": "examples/matlab/ex3_planecloud.m:18-33,79-86",
        "D1": DD1.tolist(), "D2": DD2.tolist(),
        "Agt_zero_based": (agt - 1).tolist(),
        "invariant": {"sigp": 0.5, "epsp": 0.5, "sign": math.radians(1.5), "epsn": 1.0},
    }

    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
