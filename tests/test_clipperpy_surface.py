"""`clipperpy` — the reference's Python surface (bindings/python/py_clipper.cpp:116-233).
CPU-side checks of names, defaults and host-only helpers; the GPU behaviour of the same
module is in tests/test_gpu_clipperpy.py."""
import numpy as np
import pytest

import clipper_amd
from clipper_amd import build


@pytest.fixture(scope="module")
def clipperpy():
    build.build_all()
    return clipper_amd.load_clipperpy()


def test_module_layout(clipperpy):
    assert isinstance(clipperpy.__version__, str)
    for name in ("invariants", "utils", "dsd", "CLIPPER", "Params", "Solution", "Rounding",
                 "MCParams", "SDPParams"):
        assert hasattr(clipperpy, name)
    inv = clipperpy.invariants
    for name in ("Invariant", "PairwiseInvariant", "EuclideanDistanceParams", "EuclideanDistance",
                 "PointNormalDistanceParams", "PointNormalDistance"):
        assert hasattr(inv, name)
    for meth in ("score_pairwise_consistency", "solve", "solve_as_maximum_clique",
                 "solve_as_msrc_sdr", "get_initial_associations", "get_selected_associations",
                 "get_solution", "get_affinity_matrix", "get_constraint_matrix", "set_matrix_data",
                 "set_parallelize"):
        assert hasattr(clipperpy.CLIPPER, meth)
    # py_clipper.cpp:127-128: the dsd submodule is (by a reference quirk) filled with the utils
    assert hasattr(clipperpy.dsd, "k2ij") and hasattr(clipperpy.utils, "create_all_to_all")
    assert clipperpy.NONZERO == clipperpy.Rounding.NONZERO   # export_values()


def test_param_defaults_match_reference(clipperpy):
    p = clipperpy.Params()    # clipper.h:27-60
    assert (p.tol_u, p.tol_F, p.tol_Fop, p.maxiniters, p.maxoliters, p.beta, p.maxlsiters, p.eps,
            p.affinityeps, p.rescale_u0, p.rounding) == (
        1e-8, 1e-9, 1e-10, 200, 1000, 0.25, 99, 1e-9, 1e-4, True, clipperpy.Rounding.DSD_HEU)
    e = clipperpy.invariants.EuclideanDistanceParams()   # euclidean_distance.h:22-27
    assert (e.sigma, e.epsilon, e.mindist) == (0.01, 0.06, 0)
    n = clipperpy.invariants.PointNormalDistanceParams()  # pointnormal_distance.h:25-31
    assert (n.sigp, n.epsp, n.sign, n.epsn) == (0.5, 0.5, 0.10, 0.35)
    assert "sigma=0.01" in repr(e)


def test_utils(clipperpy):
    A = clipperpy.utils.create_all_to_all(4, 3)     # utils.h:61-71
    assert A.shape == (12, 2) and A.dtype == np.int32
    assert [tuple(r) for r in A[:4]] == [(0, 0), (0, 1), (0, 2), (1, 0)]
    n = 9
    want = [(i, j) for i in range(n) for j in range(i + 1, n)]
    assert [tuple(clipperpy.utils.k2ij(k, n)) for k in range(n * (n - 1) // 2)] == want


def test_builtin_invariant_functor_on_host(clipperpy):
    # PairwiseInvariant.__call__ (py_clipper.cpp:41): euclidean_distance.cpp:13-31
    inv = clipperpy.invariants.EuclideanDistance(clipperpy.invariants.EuclideanDistanceParams())
    z, e1, e2 = np.zeros(3), np.array([1.0, 0, 0]), np.array([1.05, 0, 0])
    assert inv(z, e1, z, e1) == 1.0
    assert abs(inv(z, e1, z, e2) - np.exp(-0.5 * 0.05**2 / 0.01**2)) < 1e-15
    assert inv(z, e1, z, np.array([1.07, 0, 0])) == 0.0


def test_python_subclass_of_pairwise_invariant(clipperpy):
    class Mine(clipperpy.invariants.PairwiseInvariant):
        def __call__(self, ai, aj, bi, bj):
            return float(abs(np.linalg.norm(ai - aj) - np.linalg.norm(bi - bj)) < 0.1)

    inv = Mine()
    assert inv(np.zeros(2), np.ones(2), np.zeros(2), np.ones(2)) == 1.0
    c = clipperpy.CLIPPER(inv, clipperpy.Params())   # constructing does not touch the GPU
    assert repr(c) == "<CLIPPER>"


def test_noconvert_rejects_wrong_dtype(clipperpy):
    inv = clipperpy.invariants.EuclideanDistance(clipperpy.invariants.EuclideanDistanceParams())
    c = clipperpy.CLIPPER(inv, clipperpy.Params())
    D = np.zeros((3, 4), dtype=np.float32)          # float32 is refused, as with pybind11/eigen.h
    with pytest.raises(TypeError):
        c.score_pairwise_consistency(D, D, np.zeros((0, 2), dtype=np.int32))
    with pytest.raises(TypeError):                  # int64 associations are refused
        c.score_pairwise_consistency(np.zeros((3, 4)), np.zeros((3, 4)),
                                     np.zeros((2, 2), dtype=np.int64))
