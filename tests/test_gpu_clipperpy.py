"""GPU tests of the reference-facing surfaces: the pybind11 module `clipperpy` and the C++
class `clipper::CLIPPER` (tests/cpp/test_facade.cpp = the reference's gtest cases)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import clipper_amd
from clipper_amd import build, synth
from oracle import clipper_ref as ref

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def clipperpy():
    return clipper_amd.load_clipperpy()


def test_reference_gtest_cases_through_the_cpp_facade(tmp_path):
    exe = str(tmp_path / "test_facade")
    libdir = os.path.join(ROOT, "clipper_amd", "lib")
    subprocess.check_call([
        "g++", "-O2", "-std=c++17", "-fopenmp", "-I", os.path.join(ROOT, "include"),
        os.path.join(ROOT, "tests", "cpp", "test_facade.cpp"),
        os.path.join(ROOT, "clipper_amd", "csrc", "host", "clipper.cpp"),
        "-L", libdir, "-lclipper_hip", f"-Wl,-rpath,{libdir}", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    sys.stdout.write(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL FACADE TESTS PASSED" in out.stdout


def test_clipperpy_golden(clipperpy, golden):
    g = golden["affinity_test"]
    model, data, Mtrue = np.array(g["model"]), np.array(g["data"]), np.array(g["Mtrue"])
    iparams = clipperpy.invariants.EuclideanDistanceParams()
    inv = clipperpy.invariants.EuclideanDistance(iparams)
    c = clipperpy.CLIPPER(inv, clipperpy.Params())
    A = clipperpy.utils.create_all_to_all(4, 3)
    c.score_pairwise_consistency(model, data, A)
    assert np.array_equal(c.get_initial_associations(), A)
    assert np.array_equal(c.get_affinity_matrix(), Mtrue)
    assert np.array_equal(c.get_constraint_matrix(), Mtrue)
    c.solve(np.ones(12) / np.sqrt(12))
    Ain = c.get_selected_associations()
    assert Ain.shape == (3, 2) and np.all(Ain[:, 0] == Ain[:, 1])
    s = c.get_solution()
    assert sorted(s.nodes) == [0, 4, 8] and abs(s.score - 3) < 1e-6 and s.u.shape == (12,)
    c.solve()   # random u0, like the reference's default
    assert len(c.get_solution().nodes) in (2, 3)
    c.solve_as_msrc_sdr()       # not built: empty result, score -1 (sdp.cpp:298-302 behaviour)
    assert c.get_solution().nodes == [] and c.get_solution().score == -1


def test_clipperpy_matches_oracle_on_synthetic_problem(clipperpy):
    p = synth.make_euclidean_problem(800, 0.9, seed=5)
    ip = clipperpy.invariants.EuclideanDistanceParams()
    ip.sigma, ip.epsilon = 0.015, 0.05
    for storage in (clipperpy.Storage.F32, clipperpy.Storage.F64):
        c = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ip), clipperpy.Params())
        c.set_storage(storage)
        c.score_pairwise_consistency(p.D1, p.D2, p.A)
        c.solve(p.u0)
        r = ref.RefClipper()
        r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
        sr = r.solve(p.u0)
        s = c.get_solution()
        assert list(s.nodes) == sr.nodes.tolist()
        assert abs(s.score - sr.score) <= 1e-6 * sr.score
        assert np.array_equal(c.get_selected_associations(), r.get_selected_associations())
        st = c.get_path_stats()
        assert st["n_passes"] <= st["n_trials"] + 3 + s.ifinal


def test_clipperpy_resident_solver_switch(clipperpy):
    """The default storage solves a problem of this size with the resident (one-launch) solver;
    set_resident_solver(False) sends it to the streaming launches: the same answer."""
    p = synth.make_euclidean_problem(600, 0.9, seed=8)
    ip = clipperpy.invariants.EuclideanDistanceParams()
    ip.sigma, ip.epsilon = 0.015, 0.05
    res = []
    for on in (True, False):
        c = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ip), clipperpy.Params())
        c.set_resident_solver(on)
        c.score_pairwise_consistency(p.D1, p.D2, p.A)
        c.solve(p.u0)
        assert c.last_solve_was_resident() == on
        res.append(c.get_solution())
    assert list(res[0].nodes) == list(res[1].nodes)
    assert abs(res[0].score - res[1].score) <= 1e-9 * res[1].score


def test_clipperpy_row_views_switch(clipperpy):
    """m = 9000 through the pybind11 module: with row views (the default) most passes stream the view,
    set_row_views(False) makes every pass stream M: the same selected list and `ifinal`."""
    p = synth.make_euclidean_problem(9000, 0.95, seed=21)
    ip = clipperpy.invariants.EuclideanDistanceParams()
    ip.sigma, ip.epsilon = 0.015, 0.05
    res, on_view = [], []
    for on in (True, False):
        c = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ip), clipperpy.Params())
        c.set_row_views(on)
        c.score_pairwise_consistency(p.D1, p.D2, p.A)
        c.solve(p.u0)
        res.append(c.get_solution())
        on_view.append(c.last_solve_passes_on_a_view())
    assert on_view[0] > 0 and on_view[1] == 0
    assert list(res[0].nodes) == list(res[1].nodes) and res[0].ifinal == res[1].ifinal
    assert abs(res[0].score - res[1].score) <= 1e-9 * res[1].score


def test_clipperpy_set_devices_shards_the_columns(clipperpy):
    """multi-GPU through the class ("keeps the CLIPPER class a plain drop-in, no launcher"): set_devices with a
    device list column-shards M inside the process — here three logical shards on device 0 — and the answer is one
    shard's; the live sub-problem's switch rides along."""
    p = synth.make_euclidean_problem(9000, 0.95, seed=21)
    ip = clipperpy.invariants.EuclideanDistanceParams()
    ip.sigma, ip.epsilon = 0.015, 0.05
    res = []
    for devices in (None, [0, 0, 0]):
        c = clipperpy.CLIPPER(clipperpy.invariants.EuclideanDistance(ip), clipperpy.Params())
        if devices:
            c.set_devices(devices)
        c.set_live_subproblem(devices is None)
        c.score_pairwise_consistency(p.D1, p.D2, p.A)
        c.solve(p.u0)
        res.append(c.get_solution())
        assert c.last_solve_passes_on_the_subproblem() == 0   # (m < 12 000: the resident launch takes the view)
    assert list(res[0].nodes) == list(res[1].nodes) and res[0].ifinal == res[1].ifinal
    assert abs(res[0].score - res[1].score) <= 1e-9 * res[1].score


def test_python_custom_invariant_is_scored_on_host_and_solved_on_gpu(clipperpy):
    # the notebook's use case (examples/python/ex4_bunny.ipynb cell 12): a Python subclass
    class PyEuclid(clipperpy.invariants.PairwiseInvariant):
        def __init__(self, sigma, epsilon):
            clipperpy.invariants.PairwiseInvariant.__init__(self)
            self.sigma, self.epsilon = sigma, epsilon

        def __call__(self, ai, aj, bi, bj):
            c = abs(np.linalg.norm(ai - aj) - np.linalg.norm(bi - bj))
            return float(np.exp(-0.5 * c * c / self.sigma**2)) if c < self.epsilon else 0.0

    p = synth.make_euclidean_problem(120, 0.8, seed=9)
    c = clipperpy.CLIPPER(PyEuclid(0.015, 0.05), clipperpy.Params())
    c.score_pairwise_consistency(p.D1, p.D2, p.A)
    r = ref.RefClipper()
    r.score_pairwise_consistency_euclidean(p.D1, p.D2, p.A, **synth.EUCLID_BENCH_PARAMS)
    Mg, Mr = c.get_affinity_matrix(), r.get_affinity_matrix()
    assert np.array_equal(Mg != 0, Mr != 0)
    assert np.allclose(Mg, Mr, rtol=0, atol=2e-7)     # fp32 storage of a numpy-computed score
    c.solve(p.u0)
    sr = r.solve(p.u0)
    assert list(c.get_solution().nodes) == sr.nodes.tolist()
    assert abs(c.get_solution().score - sr.score) <= 1e-6 * sr.score


def test_clipperpy_set_matrix_data(clipperpy, golden):
    M = np.array(golden["dsd_test_20x20"]["M"])
    Cm = (M > 0).astype(float)
    inv = clipperpy.invariants.EuclideanDistance(clipperpy.invariants.EuclideanDistanceParams())
    c = clipperpy.CLIPPER(inv, clipperpy.Params())
    c.set_storage(clipperpy.Storage.F64)
    c.set_matrix_data(M, Cm)
    assert np.array_equal(c.get_affinity_matrix(), M)
    r = ref.RefClipper()
    r.set_matrix_data(M, Cm)
    u0 = np.ones(20) / np.sqrt(20)
    c.solve(u0)
    sr = r.solve(u0)
    assert list(c.get_solution().nodes) == sr.nodes.tolist()
