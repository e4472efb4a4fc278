"""ctypes front-end of the CPU ORACLE (oracle/libclipper_ref.so) plus an independent
numpy restatement used to cross-check it.

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import this module (see oracle/clipper_ref.cpp header).

Two restatements of the same reference code are kept on purpose:
  * `RefClipper`            -> the C++ oracle (fast; also the timed CPU baseline),
  * `numpy_affinity_*`, `numpy_solve` -> dense numpy mirror, written separately from the
    C++ one; tests require the two to agree, which guards against a transcription slip
    in either.
Reference citations are relative to /root/reference.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libclipper_ref.so")


class Params(C.Structure):
    """clipper_params_t (include/clipper_abi.h) == clipper::Params (clipper.h:27-60)."""

    _fields_ = [
        ("tol_u", C.c_double),
        ("tol_F", C.c_double),
        ("tol_Fop", C.c_double),
        ("maxiniters", C.c_int32),
        ("maxoliters", C.c_int32),
        ("beta", C.c_double),
        ("maxlsiters", C.c_int32),
        ("eps", C.c_double),
        ("affinityeps", C.c_double),
        ("rescale_u0", C.c_int32),
        ("rounding", C.c_int32),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.tol_u, self.tol_F, self.tol_Fop = 1e-8, 1e-9, 1e-10
        self.maxiniters, self.maxoliters = 200, 1000
        self.beta, self.maxlsiters = 0.25, 99
        self.eps, self.affinityeps = 1e-9, 1e-4
        self.rescale_u0, self.rounding = 1, 2
        for k, v in kw.items():
            setattr(self, k, v)


class SolveInfo(C.Structure):
    """clipper_solve_info_t (include/clipper_abi.h)."""

    _fields_ = [
        ("score", C.c_double),
        ("seconds", C.c_double),
        ("d", C.c_double),
        ("ifinal", C.c_int32),
        ("num_nodes", C.c_int32),
        ("n_passes", C.c_int64),
        ("n_trials", C.c_int64),
    ]


ROUNDING_NONZERO, ROUNDING_DSD, ROUNDING_DSD_HEU = 0, 1, 2


def build(force: bool = False) -> str:
    """Compile oracle/libclipper_ref.so with the committed Makefile (g++ only)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "clipper_ref.cpp"))
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(_LIB_PATH)
    dp, ip, i64 = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64
    L.clipper_ref_create.restype = C.c_void_p
    L.clipper_ref_destroy.argtypes = [C.c_void_p]
    L.clipper_ref_last_error.restype = C.c_char_p
    L.clipper_ref_affinity_euclidean.argtypes = [
        C.c_void_p, dp, C.c_int, i64, dp, i64, ip, i64,
        C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    L.clipper_ref_affinity_pointnormal.argtypes = [
        C.c_void_p, dp, C.c_int, i64, dp, i64, ip, i64,
        C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.c_int]
    L.clipper_ref_num_associations.argtypes = [C.c_void_p]
    L.clipper_ref_num_associations.restype = i64
    L.clipper_ref_get_associations.argtypes = [C.c_void_p, ip]
    L.clipper_ref_set_matrix.argtypes = [C.c_void_p, dp, dp, i64]
    L.clipper_ref_set_sparse.argtypes = [
        C.c_void_p, i64, C.POINTER(i64), ip, dp, C.POINTER(i64), ip, dp]
    L.clipper_ref_get_matrix.argtypes = [C.c_void_p, dp, dp]
    L.clipper_ref_nnz.argtypes = [C.c_void_p]
    L.clipper_ref_set_sum_mode.argtypes = [C.c_void_p, C.c_int]
    L.clipper_ref_nnz.restype = i64
    L.clipper_ref_solve.argtypes = [C.c_void_p, dp, C.POINTER(Params), dp, C.POINTER(SolveInfo)]
    L.clipper_ref_get_nodes.argtypes = [C.c_void_p, ip, C.c_int32]
    L.clipper_ref_get_selected_associations.argtypes = [C.c_void_p, ip, C.c_int32]
    L.clipper_ref_matvec.argtypes = [C.c_void_p, dp, dp, dp]
    L.clipper_ref_k2ij.argtypes = [C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64),
                                   C.POINTER(C.c_uint64)]
    L.clipper_ref_k2ij.restype = None
    L.clipper_ref_create_all_to_all.argtypes = [i64, i64, ip]
    L.clipper_ref_create_all_to_all.restype = None
    L.clipper_ref_k_largest.argtypes = [dp, i64, C.c_int32, ip]
    L.clipper_ref_score_euclidean.argtypes = [dp, dp, dp, dp, C.c_int, C.c_double, C.c_double,
                                              C.c_double]
    L.clipper_ref_score_euclidean.restype = C.c_double
    L.clipper_ref_score_pointnormal.argtypes = [dp, dp, dp, dp, C.c_double, C.c_double,
                                                C.c_double, C.c_double]
    L.clipper_ref_score_pointnormal.restype = C.c_double
    L.clipper_ref_omp_threads.restype = C.c_int
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _f64_colmajor(D):
    """d x n array -> contiguous column-major buffer (each datum contiguous)."""
    return np.asfortranarray(np.asarray(D, dtype=np.float64))


def _assoc_colmajor(A):
    """m x 2 int array -> column-major int32 buffer (Eigen::Matrix<int,Dynamic,2>)."""
    A = np.asarray(A)
    if A.size == 0:
        return None, 0
    A = np.asfortranarray(A.astype(np.int32, copy=False))
    assert A.ndim == 2 and A.shape[1] == 2
    return A, A.shape[0]


@dataclass
class Solution:
    """clipper::Solution (clipper.h:65-73) + pass counters."""

    t: float = 0.0
    ifinal: int = 0
    nodes: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    u0: np.ndarray = field(default_factory=lambda: np.zeros(0))
    u: np.ndarray = field(default_factory=lambda: np.zeros(0))
    score: float = 0.0
    d: float = 0.0
    n_passes: int = 0
    n_trials: int = 0


class RefClipper:
    """Python handle on the C++ oracle; method names follow clipperpy (py_clipper.cpp:197-232)."""

    def __init__(self, params: Params | None = None):
        self.L = lib()
        self.h = C.c_void_p(self.L.clipper_ref_create())
        self.params = params or Params()
        self.parallelize = True
        self.soln = Solution()

    def __del__(self):
        try:
            if self.h:
                self.L.clipper_ref_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise RuntimeError(self.L.clipper_ref_last_error().decode())

    def score_pairwise_consistency_euclidean(self, D1, D2, A=(), sigma=0.01, epsilon=0.06,
                                             mindist=0.0, dense_temp=-1):
        D1, D2 = _f64_colmajor(D1), _f64_colmajor(D2)
        Ac, m = _assoc_colmajor(A)
        self._check(self.L.clipper_ref_affinity_euclidean(
            self.h, _dp(D1), D1.shape[0], D1.shape[1], _dp(D2), D2.shape[1],
            _ip(Ac) if Ac is not None else None, m, sigma, epsilon, mindist,
            self.params.affinityeps, int(self.parallelize), dense_temp))

    def score_pairwise_consistency_pointnormal(self, D1, D2, A=(), sigp=0.5, epsp=0.5, sign=0.10,
                                               epsn=0.35, dense_temp=-1):
        D1, D2 = _f64_colmajor(D1), _f64_colmajor(D2)
        Ac, m = _assoc_colmajor(A)
        self._check(self.L.clipper_ref_affinity_pointnormal(
            self.h, _dp(D1), D1.shape[0], D1.shape[1], _dp(D2), D2.shape[1],
            _ip(Ac) if Ac is not None else None, m, sigp, epsp, sign, epsn,
            self.params.affinityeps, int(self.parallelize), dense_temp))

    @property
    def m(self):
        return int(self.L.clipper_ref_num_associations(self.h))

    @property
    def nnz(self):
        return int(self.L.clipper_ref_nnz(self.h))

    def set_sum_mode(self, mode: int):
        """order of the additions inside M x, C x: 0 = the reference's (parity), 1 = reversed, 2 = extended precision"""
        self._check(self.L.clipper_ref_set_sum_mode(self.h, int(mode)))

    def get_initial_associations(self):
        A = np.zeros((self.m, 2), dtype=np.int32, order="F")
        self._check(self.L.clipper_ref_get_associations(self.h, _ip(A)))
        return np.ascontiguousarray(A)

    def get_affinity_matrix(self):
        m = self.m
        M = np.zeros((m, m), order="F")
        self._check(self.L.clipper_ref_get_matrix(self.h, _dp(M), None))
        return M

    def get_constraint_matrix(self):
        m = self.m
        Cm = np.zeros((m, m), order="F")
        self._check(self.L.clipper_ref_get_matrix(self.h, None, _dp(Cm)))
        return Cm

    def set_matrix_data(self, M, Cm):
        M, Cm = _f64_colmajor(M), _f64_colmajor(Cm)
        self._check(self.L.clipper_ref_set_matrix(self.h, _dp(M), _dp(Cm), M.shape[0]))

    def set_sparse_matrix_data(self, m, Mcolptr, Mrow, Mval, Ccolptr, Crow, Cval):
        """CLIPPER::setSparseMatrixData (clipper.cpp:162-166): the CSC arrays are kept as handed over;
        every product reads them through selfadjointView<Upper>."""
        i64p = C.POINTER(C.c_int64)
        a = [np.ascontiguousarray(Mcolptr, np.int64), np.ascontiguousarray(Mrow, np.int32),
             np.ascontiguousarray(Mval, np.float64), np.ascontiguousarray(Ccolptr, np.int64),
             np.ascontiguousarray(Crow, np.int32), np.ascontiguousarray(Cval, np.float64)]
        self._check(self.L.clipper_ref_set_sparse(
            self.h, int(m), a[0].ctypes.data_as(i64p), _ip(a[1]), _dp(a[2]),
            a[3].ctypes.data_as(i64p), _ip(a[4]), _dp(a[5])))

    def matvec(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        yM, yC = np.zeros_like(x), np.zeros_like(x)
        self._check(self.L.clipper_ref_matvec(self.h, _dp(x), _dp(yM), _dp(yC)))
        return yM, yC

    def solve(self, u0):
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        n = self.m
        assert u0.shape == (n,)
        u = np.zeros(n)
        info = SolveInfo()
        self._check(self.L.clipper_ref_solve(self.h, _dp(u0), C.byref(self.params), _dp(u),
                                             C.byref(info)))
        nodes = np.zeros(max(info.num_nodes, 1), dtype=np.int32)
        k = self.L.clipper_ref_get_nodes(self.h, _ip(nodes), nodes.size)
        self.soln = Solution(t=info.seconds, ifinal=info.ifinal, nodes=nodes[:k].copy(), u0=u0,
                             u=u, score=info.score, d=info.d, n_passes=info.n_passes,
                             n_trials=info.n_trials)
        return self.soln

    def get_solution(self):
        return self.soln

    def get_selected_associations(self):
        return _unpack_kx2(self.L, self.h, len(self.soln.nodes))


def _unpack_kx2(L, h, k):
    buf = np.zeros(2 * max(k, 1), dtype=np.int32)
    kk = L.clipper_ref_get_selected_associations(h, _ip(buf), max(k, 1))
    if kk <= 0:
        return np.zeros((0, 2), dtype=np.int32)
    return np.stack([buf[:kk], buf[kk:2 * kk]], axis=1)


# --------------------------------------------------------------------------------------
# utils
# --------------------------------------------------------------------------------------

def k2ij(k: int, n: int):
    """utils::k2ij (utils.cpp:87-97)."""
    i, j = C.c_uint64(), C.c_uint64()
    lib().clipper_ref_k2ij(k, n, C.byref(i), C.byref(j))
    return int(i.value), int(j.value)


def create_all_to_all(n1: int, n2: int):
    """utils::createAllToAll (utils.h:61-71)."""
    A = np.zeros((n1 * n2, 2), dtype=np.int32, order="F")
    lib().clipper_ref_create_all_to_all(n1, n2, _ip(A))
    return np.ascontiguousarray(A)


def k_largest(x, k: int):
    """utils::findIndicesOfkLargest (utils.cpp:33-55)."""
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros(max(min(k, x.size), 1), dtype=np.int32)
    r = lib().clipper_ref_k_largest(_dp(x), x.size, k, _ip(out))
    return out[:r].copy()


def omp_threads() -> int:
    return int(lib().clipper_ref_omp_threads())


# --------------------------------------------------------------------------------------
# independent numpy mirror (dense), for cross-checking the C++ oracle on small cases
# --------------------------------------------------------------------------------------

def numpy_affinity_euclidean(D1, D2, A, sigma=0.01, epsilon=0.06, mindist=0.0, affinityeps=1e-4):
    """clipper.cpp:21-65 with euclidean_distance.cpp:13-31; returns dense strict-upper M."""
    D1, D2, A = np.asarray(D1, float), np.asarray(D2, float), np.asarray(A)
    m = A.shape[0]
    P1, P2 = D1[:, A[:, 0]].T, D2[:, A[:, 1]].T  # m x d
    l1 = np.sqrt(((P1[:, None, :] - P1[None, :, :]) ** 2).sum(-1))
    l2 = np.sqrt(((P2[:, None, :] - P2[None, :, :]) ** 2).sum(-1))
    c = np.abs(l1 - l2)
    with np.errstate(over="ignore"):
        scr = np.where(c < epsilon, np.exp(-0.5 * c * c / (sigma * sigma)), 0.0)
    if mindist > 0:
        scr = np.where((l1 < mindist) | (l2 < mindist), 0.0, scr)
    distinct = (A[:, None, 0] != A[None, :, 0]) & (A[:, None, 1] != A[None, :, 1])
    M = np.where(distinct & (scr > affinityeps), scr, 0.0)
    return np.triu(M, 1), m


def numpy_affinity_pointnormal(D1, D2, A, sigp=0.5, epsp=0.5, sign=0.10, epsn=0.35,
                               affinityeps=1e-4):
    """clipper.cpp:21-65 with pointnormal_distance.cpp:13-35; returns dense strict-upper M."""
    D1, D2, A = np.asarray(D1, float), np.asarray(D2, float), np.asarray(A)
    P1, P2 = D1[:, A[:, 0]].T, D2[:, A[:, 1]].T  # m x 6
    l1 = np.sqrt(((P1[:, None, :3] - P1[None, :, :3]) ** 2).sum(-1))
    l2 = np.sqrt(((P2[:, None, :3] - P2[None, :, :3]) ** 2).sum(-1))
    with np.errstate(invalid="ignore"):
        a1 = np.arccos(P1[:, 3:] @ P1[:, 3:].T)
        a2 = np.arccos(P2[:, 3:] @ P2[:, 3:].T)
        dp, dn = np.abs(l1 - l2), np.abs(a1 - a2)
        ok = (dp < epsp) & (dn < epsn)  # NaN compares false, as in the reference
        scr = np.where(ok, np.exp(-0.5 * dp * dp / sigp**2) * np.exp(-0.5 * dn * dn / sign**2), 0.0)
    distinct = (A[:, None, 0] != A[None, :, 0]) & (A[:, None, 1] != A[None, :, 1])
    M = np.where(distinct & (scr > affinityeps), scr, 0.0)
    return np.triu(M, 1)


def numpy_k_largest(x, k):
    """utils.cpp:33-55 with a heap, in pure Python (small inputs only)."""
    import heapq
    if k < 1:
        return np.zeros(0, np.int32)
    k = min(k, len(x))
    q = []
    for i, v in enumerate(x):
        if len(q) < k:
            heapq.heappush(q, (float(v), i))
        elif q[0][0] < v:
            heapq.heapreplace(q, (float(v), i))
    out = [0] * k
    for i in range(k):
        out[k - i - 1] = heapq.heappop(q)[1]
    return np.array(out, dtype=np.int32)


class _Op:
    """`op @ x` through a callable, so that numpy_solve can run on a distributed mat-vec."""

    def __init__(self, fn):
        self.fn = fn

    def __matmul__(self, x):
        return self.fn(x)


def numpy_solve(Mup, Cup, u0, p: Params | None = None, matvec=None):
    """findDenseClique (clipper.cpp:172-323) on dense strict-upper M, C (numpy, fp64).
    `matvec(x) -> (M_off x, C_off x)`, when given, replaces the dense products (used by the
    world_size-2 gloo test to run the same loop on a column-sharded, all-gathered mat-vec)."""
    p = p or Params()
    if matvec is None:
        Ms = Mup + Mup.T  # selfadjointView<Upper> of the strict upper part
        Cs = Cup + Cup.T
    else:
        Ms, Cs = _Op(lambda x: matvec(x)[0]), _Op(lambda x: matvec(x)[1])
    u0 = np.asarray(u0, float)
    n = u0.shape[0]
    u = Ms @ u0 + u0 if p.rescale_u0 else u0.copy()
    u = u / np.linalg.norm(u)
    d = 0.0
    Cbu = u.sum() - Cs @ u - u
    idx = (Cbu > p.eps) & (u > p.eps)
    if idx.sum() > 0:
        Mu = Ms @ u + u
        d = float(np.mean(Mu[idx] / Cbu[idx]))
    F = 0.0
    n_trials = 0
    i = 0
    for i in range(p.maxoliters + 1):
        if i == p.maxoliters:
            break
        gradF = (1 + d) * u - d * u.sum() + Ms @ u + Cs @ u * d
        F = float(u @ gradF)
        for _j in range(p.maxiniters):
            alpha = 1.0
            Fnew = deltaF = 0.0
            for _k in range(p.maxlsiters):
                unew = np.maximum(u + alpha * gradF, 0)
                z = float(unew @ unew)
                if z > 0:
                    unew = unew / np.sqrt(z)
                gradFnew = (1 + d) * unew - d * unew.sum() + Ms @ unew + Cs @ unew * d
                n_trials += 1
                Fnew = float(unew @ gradFnew)
                deltaF = Fnew - F
                if deltaF < -p.eps:
                    alpha *= p.beta
                else:
                    break
            deltau = float(np.linalg.norm(unew - u))
            F, u, gradF = Fnew, unew, gradFnew
            if deltau < p.tol_u or abs(deltaF) < p.tol_F:
                break
        Cbu = u.sum() - Cs @ u - u
        idx = (Cbu > p.eps) & (u > p.eps)
        if idx.sum() > 0:
            Mu = Ms @ u + u
            d += float(np.mean(np.abs(Mu[idx] / Cbu[idx])))
        else:
            break
    if p.rounding == ROUNDING_NONZERO:
        nodes = np.nonzero(u > 0)[0].astype(np.int32)
    elif p.rounding == ROUNDING_DSD:
        # clipper.cpp:294-300: exact densest subgraph of the graph induced by nnz(u)
        from oracle import dsd_ref
        if matvec is not None:
            raise ValueError("Rounding::DSD needs the matrix itself")
        S = np.nonzero(u > 0)[0].tolist()
        nodes = np.array(dsd_ref.densest_subgraph(Mup + Mup.T, S), dtype=np.int32)
    else:
        nodes = numpy_k_largest(u, int(np.floor(F + 0.5)) if F >= 0 else -int(np.floor(-F + 0.5)))
    return Solution(ifinal=i, nodes=nodes, u0=u0, u=u, score=F, d=d, n_trials=n_trials)
