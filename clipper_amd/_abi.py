"""ctypes binding of the C ABI in include/clipper_hip.h (clipper_amd/lib/libclipper_hip.so).

This is the thinnest possible Python view of the drop-in boundary: every method is one
C call. The user-facing Python surface of the reference (`clipperpy`, py_clipper.cpp) is
provided by the pybind11 module built from clipper_amd/csrc/host/; this module is what the
parity tests and bench.py drive so that they exercise exactly the exported symbols.

There is no fallback: if the shared library is missing or no HIP device is usable, the
constructors raise.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CLIPPER_HIP_LIB: alternate build of the same library (A/B measurements of kernel variants)
LIB_PATH = os.environ.get("CLIPPER_HIP_LIB") or os.path.join(_HERE, "lib", "libclipper_hip.so")

STORE_F32, STORE_F64, STORE_F32_CSC, STORE_F64_CSC = 0, 1, 2, 3
ROUNDING_NONZERO, ROUNDING_DSD, ROUNDING_DSD_HEU = 0, 1, 2

# every symbol include/clipper_hip.h declares (checked by tests/test_abi_exports.py)
EXPORTED_SYMBOLS = [
    "clipper_hip_device_count", "clipper_hip_create", "clipper_hip_create_group",
    "clipper_hip_create_rank", "clipper_hip_comm_unique_id", "clipper_hip_comm_init",
    "clipper_hip_destroy", "clipper_hip_last_error", "clipper_hip_affinity_euclidean",
    "clipper_hip_affinity_pointnormal", "clipper_hip_num_associations",
    "clipper_hip_get_associations", "clipper_hip_set_matrix", "clipper_hip_set_sparse",
    "clipper_hip_get_matrix", "clipper_hip_solve", "clipper_hip_get_nodes",
    "clipper_hip_get_selected_associations", "clipper_hip_matvec", "clipper_hip_set_profiling",
    "clipper_hip_set_window", "clipper_hip_window", "clipper_hip_densest_subgraph",
    "clipper_hip_set_resident", "clipper_hip_last_solver",
    "clipper_hip_set_row_view", "clipper_hip_get_view_stats", "clipper_hip_view_matvec", "clipper_hip_set_subproblem",
    "clipper_hip_storage_in_use", "clipper_hip_knn", "clipper_hip_distance_based_correspondences",
    "clipper_hip_get_timings", "clipper_hip_bench_matvec", "clipper_hip_device_info",
    "clipper_hip_stage_inputs", "clipper_hip_affinity_euclidean_staged",
    "clipper_hip_affinity_pointnormal_staged", "clipper_hip_stage_u0",
    "clipper_hip_solve_staged", "clipper_hip_debug_stamps", "clipper_hip_comm_init_callback",
    "clipper_hip_read_ply_xyz", "clipper_hip_generate_synthetic_correspondences",
    "clipper_hip_precision_recall", "clipper_hip_estimate_rigid_transform", "clipper_hip_debug_occupy",
]


# int fn(void* user, const void* sendbuf, void* recvbuf, size_t bytes) — clipper_hip_allgather_fn
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)


class ClipperError(RuntimeError):
    """a C ABI entry point returned a negative status (CLIPPER_HIP_E_*)"""


class Params(C.Structure):
    """clipper_params_t == clipper::Params (reference include/clipper/clipper.h:27-60)."""

    _fields_ = [
        ("tol_u", C.c_double), ("tol_F", C.c_double), ("tol_Fop", C.c_double),
        ("maxiniters", C.c_int32), ("maxoliters", C.c_int32), ("beta", C.c_double),
        ("maxlsiters", C.c_int32), ("eps", C.c_double), ("affinityeps", C.c_double),
        ("rescale_u0", C.c_int32), ("rounding", C.c_int32),
    ]

    def __init__(self, **kw):
        super().__init__()
        self.tol_u, self.tol_F, self.tol_Fop = 1e-8, 1e-9, 1e-10
        self.maxiniters, self.maxoliters = 200, 1000
        self.beta, self.maxlsiters = 0.25, 99
        self.eps, self.affinityeps = 1e-9, 1e-4
        self.rescale_u0, self.rounding = 1, ROUNDING_DSD_HEU
        for k, v in kw.items():
            setattr(self, k, v)


class SolveInfo(C.Structure):
    _fields_ = [
        ("score", C.c_double), ("seconds", C.c_double), ("d", C.c_double),
        ("ifinal", C.c_int32), ("num_nodes", C.c_int32),
        ("n_passes", C.c_int64), ("n_trials", C.c_int64),
    ]


class Timings(C.Structure):
    _fields_ = [
        ("affinity_kernel_ms", C.c_double), ("affinity_total_ms", C.c_double),
        ("solve_total_ms", C.c_double), ("gemv_avg_us", C.c_double),
        ("gemv_min_us", C.c_double), ("gemv_launches", C.c_int64), ("gemv_bytes", C.c_double),
        ("gemv_useful_bytes", C.c_double), ("affinity_bytes", C.c_double),
        ("exchange_avg_us", C.c_double), ("exchange_samples", C.c_int64), ("exchange_bytes", C.c_double),
    ]


class ViewStats(C.Structure):
    """clipper_hip_view_stats_t (include/clipper_hip.h): the row views of the last solve."""

    _fields_ = [
        ("builds", C.c_int64), ("rows", C.c_int64), ("bytes", C.c_int64),
        ("view_passes", C.c_int64), ("passes", C.c_int64), ("build_ms", C.c_double),
        ("view_pass_avg_us", C.c_double), ("view_pass_samples", C.c_int64),
        ("resident_launches", C.c_int64), ("resident_giveups", C.c_int64),
        ("resident_iterations", C.c_int64), ("resident_us", C.c_double), ("resident_event_us", C.c_double),
        ("resident_entries", C.c_int64), ("resident_units", C.c_int64),
        ("sub_entries", C.c_int64), ("sub_leaves", C.c_int64), ("sub_passes", C.c_int64),
        ("sub_rows", C.c_int64), ("sub_bytes", C.c_int64), ("sub_build_ms", C.c_double),
        ("sub_pass_avg_us", C.c_double), ("sub_pass_samples", C.c_int64), ("sub_dense", C.c_int64),
    ]


@dataclass
class Solution:
    """clipper::Solution (clipper.h:65-73) plus the counters this build reports."""

    t: float = 0.0
    ifinal: int = 0
    nodes: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    u0: np.ndarray = field(default_factory=lambda: np.zeros(0))
    u: np.ndarray = field(default_factory=lambda: np.zeros(0))
    score: float = 0.0
    d: float = 0.0
    n_passes: int = 0
    n_trials: int = 0


_lib = None


def load_library(path: str = LIB_PATH):
    """dlopen the product library and declare the prototypes. Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'`"
            " (hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = C.CDLL(path)
    dp, ip, i64, vp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int64, C.c_void_p
    L.clipper_hip_device_count.restype = C.c_int
    L.clipper_hip_create.argtypes = [C.c_int, C.c_int]
    L.clipper_hip_create.restype = vp
    L.clipper_hip_create_group.argtypes = [C.POINTER(C.c_int), C.c_int, C.c_int]
    L.clipper_hip_create_group.restype = vp
    L.clipper_hip_create_rank.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    L.clipper_hip_create_rank.restype = vp
    L.clipper_hip_comm_unique_id.argtypes = [vp]
    L.clipper_hip_comm_init.argtypes = [vp, vp]
    L.clipper_hip_destroy.argtypes = [vp]
    L.clipper_hip_destroy.restype = None
    L.clipper_hip_last_error.restype = C.c_char_p
    L.clipper_hip_affinity_euclidean.argtypes = [
        vp, dp, C.c_int, i64, dp, i64, ip, i64, C.c_double, C.c_double, C.c_double, C.c_double]
    L.clipper_hip_affinity_pointnormal.argtypes = [
        vp, dp, C.c_int, i64, dp, i64, ip, i64, C.c_double, C.c_double, C.c_double, C.c_double,
        C.c_double]
    L.clipper_hip_stage_inputs.argtypes = [vp, dp, C.c_int, i64, dp, i64, ip, i64]
    L.clipper_hip_affinity_euclidean_staged.argtypes = [vp] + [C.c_double] * 4
    L.clipper_hip_affinity_pointnormal_staged.argtypes = [vp] + [C.c_double] * 5
    L.clipper_hip_stage_u0.argtypes = [vp, dp]
    L.clipper_hip_solve_staged.argtypes = [vp, C.POINTER(Params), dp, C.POINTER(SolveInfo)]
    L.clipper_hip_num_associations.argtypes = [vp]
    L.clipper_hip_num_associations.restype = i64
    L.clipper_hip_get_associations.argtypes = [vp, ip]
    L.clipper_hip_set_matrix.argtypes = [vp, dp, dp, i64]
    L.clipper_hip_set_sparse.argtypes = [vp, i64, C.POINTER(i64), ip, dp, C.POINTER(i64), ip, dp]
    L.clipper_hip_get_matrix.argtypes = [vp, dp, dp]
    L.clipper_hip_solve.argtypes = [vp, dp, C.POINTER(Params), dp, C.POINTER(SolveInfo)]
    L.clipper_hip_get_nodes.argtypes = [vp, ip, C.c_int32]
    L.clipper_hip_get_selected_associations.argtypes = [vp, ip, C.c_int32]
    L.clipper_hip_matvec.argtypes = [vp, dp, dp, dp]
    L.clipper_hip_densest_subgraph.argtypes = [vp, ip, C.c_int32, ip, C.c_int32]
    L.clipper_hip_set_window.argtypes = [vp, C.c_int]
    L.clipper_hip_window.argtypes = [vp]
    L.clipper_hip_set_resident.argtypes = [vp, C.c_int]
    L.clipper_hip_last_solver.argtypes = [vp]
    L.clipper_hip_storage_in_use.argtypes = [vp]
    L.clipper_hip_set_row_view.argtypes = [vp, C.c_int]
    L.clipper_hip_set_subproblem.argtypes = [vp, C.c_int]
    L.clipper_hip_get_view_stats.argtypes = [vp, C.POINTER(ViewStats)]
    L.clipper_hip_view_matvec.argtypes = [vp, ip, i64, dp, dp, dp]
    L.clipper_hip_knn.argtypes = [C.c_int, dp, C.c_int64, dp, C.c_int64, C.c_int, C.c_int, ip, dp]
    L.clipper_hip_distance_based_correspondences.argtypes = [
        C.c_int, dp, C.c_int64, dp, C.c_int64, C.c_int, C.c_int, C.c_double, C.c_int, ip, C.c_int64]
    L.clipper_hip_distance_based_correspondences.restype = C.c_int64
    L.clipper_hip_set_profiling.argtypes = [vp, C.c_int]
    L.clipper_hip_get_timings.argtypes = [vp, C.POINTER(Timings)]
    L.clipper_hip_bench_matvec.argtypes = [vp, C.c_int, dp]
    L.clipper_hip_device_info.argtypes = [vp, C.c_char_p, C.POINTER(C.c_int), C.POINTER(i64)]
    L.clipper_hip_debug_stamps.argtypes = [vp, C.POINTER(i64), C.c_int]
    L.clipper_hip_comm_init_callback.argtypes = [vp, ALLGATHER_FN, vp]
    L.clipper_hip_read_ply_xyz.argtypes = [C.c_char_p, dp, i64]
    L.clipper_hip_read_ply_xyz.restype = i64
    L.clipper_hip_generate_synthetic_correspondences.argtypes = [i64, i64, ip, i64, i64, C.c_double, C.c_uint64,
                                                                 ip, ip, C.POINTER(i64)]
    L.clipper_hip_precision_recall.argtypes = [ip, i64, ip, i64, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.clipper_hip_estimate_rigid_transform.argtypes = [dp, i64, dp, i64, ip, i64, dp]
    L.clipper_hip_debug_occupy.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double]
    _lib = L
    return L


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def _i64p(a):
    return a.ctypes.data_as(C.POINTER(C.c_int64))


def _f64_colmajor(D):
    return np.asfortranarray(np.asarray(D, dtype=np.float64))


def _assoc_colmajor(A):
    A = np.asarray(A)
    if A.size == 0:
        return None, 0
    A = np.asfortranarray(A.astype(np.int32, copy=False))
    if A.ndim != 2 or A.shape[1] != 2:
        raise ValueError("A must be m x 2")
    return A, A.shape[0]


class HipClipper:
    """One problem instance on the GPU(s); method names follow clipperpy.CLIPPER."""

    def __init__(self, params: Params | None = None, device: int = 0, storage: int = STORE_F32,
                 group: list[int] | None = None, rank: int | None = None, world: int = 1):
        self.L = load_library()
        self.params = params or Params()
        if group is not None:
            arr = (C.c_int * len(group))(*group)
            h = self.L.clipper_hip_create_group(arr, len(group), storage)
        elif rank is not None:
            h = self.L.clipper_hip_create_rank(device, storage, rank, world)
        else:
            h = self.L.clipper_hip_create(device, storage)
        if not h:
            raise RuntimeError("clipper_hip_create failed: " + self.last_error())
        self.h = C.c_void_p(h)
        self.storage = storage
        self.soln = Solution()

    def last_error(self) -> str:
        return self.L.clipper_hip_last_error().decode()

    def close(self):
        if getattr(self, "h", None):
            self.L.clipper_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):  # raises ClipperError (a RuntimeError) for a negative status
        if rc < 0:
            raise ClipperError(f"clipper_hip error {rc}: {self.last_error()}")
        return rc

    # ---- communicator (multi-process shards) ----------------------------------------------
    def unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self._check(self.L.clipper_hip_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes):
        buf = C.create_string_buffer(uid, 128)
        self._check(self.L.clipper_hip_comm_init(self.h, buf))

    # ---- affinity --------------------------------------------------------------------------
    def score_pairwise_consistency_euclidean(self, D1, D2, A=(), sigma=0.01, epsilon=0.06,
                                             mindist=0.0):
        D1, D2 = _f64_colmajor(D1), _f64_colmajor(D2)
        if D1.shape[0] != D2.shape[0]:
            raise ValueError("D1 and D2 must have the same number of rows")
        Ac, m = _assoc_colmajor(A)
        self._check(self.L.clipper_hip_affinity_euclidean(
            self.h, _dp(D1), D1.shape[0], D1.shape[1], _dp(D2), D2.shape[1],
            _ip(Ac) if Ac is not None else None, m, sigma, epsilon, mindist,
            self.params.affinityeps))

    def score_pairwise_consistency_pointnormal(self, D1, D2, A=(), sigp=0.5, epsp=0.5, sign=0.10,
                                               epsn=0.35):
        D1, D2 = _f64_colmajor(D1), _f64_colmajor(D2)
        if D1.shape[0] != 6 or D2.shape[0] != 6:
            raise ValueError("PointNormalDistance data are 6 x n (xyz + unit normal)")
        Ac, m = _assoc_colmajor(A)
        self._check(self.L.clipper_hip_affinity_pointnormal(
            self.h, _dp(D1), D1.shape[0], D1.shape[1], _dp(D2), D2.shape[1],
            _ip(Ac) if Ac is not None else None, m, sigp, epsp, sign, epsn,
            self.params.affinityeps))

    # split forms: inputs resident in HBM, device work callable (and timeable) on its own
    def stage_inputs(self, D1, D2, A=()):
        D1, D2 = _f64_colmajor(D1), _f64_colmajor(D2)
        if D1.shape[0] != D2.shape[0]:
            raise ValueError("D1 and D2 must have the same number of rows")
        Ac, m = _assoc_colmajor(A)
        self._check(self.L.clipper_hip_stage_inputs(
            self.h, _dp(D1), D1.shape[0], D1.shape[1], _dp(D2), D2.shape[1],
            _ip(Ac) if Ac is not None else None, m))

    def affinity_euclidean_staged(self, sigma=0.01, epsilon=0.06, mindist=0.0):
        self._check(self.L.clipper_hip_affinity_euclidean_staged(
            self.h, sigma, epsilon, mindist, self.params.affinityeps))

    def affinity_pointnormal_staged(self, sigp=0.5, epsp=0.5, sign=0.10, epsn=0.35):
        self._check(self.L.clipper_hip_affinity_pointnormal_staged(
            self.h, sigp, epsp, sign, epsn, self.params.affinityeps))

    def stage_u0(self, u0):
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        if u0.shape != (self.m,):
            raise ValueError(f"u0 must have shape ({self.m},)")
        self._u0 = u0
        self._check(self.L.clipper_hip_stage_u0(self.h, _dp(u0)))

    def solve_staged(self):
        n = self.m
        u = np.zeros(n)
        info = SolveInfo()
        self._check(self.L.clipper_hip_solve_staged(self.h, C.byref(self.params), _dp(u),
                                                    C.byref(info)))
        nodes = np.zeros(max(info.num_nodes, 1), dtype=np.int32)
        k = self._check(self.L.clipper_hip_get_nodes(self.h, _ip(nodes), nodes.size))
        self.soln = Solution(t=info.seconds, ifinal=info.ifinal, nodes=nodes[:k].copy(),
                             u0=getattr(self, "_u0", np.zeros(0)), u=u, score=info.score,
                             d=info.d, n_passes=info.n_passes, n_trials=info.n_trials)
        return self.soln

    @property
    def m(self) -> int:
        return int(self.L.clipper_hip_num_associations(self.h))

    def get_initial_associations(self):
        A = np.zeros((self.m, 2), dtype=np.int32, order="F")
        self._check(self.L.clipper_hip_get_associations(self.h, _ip(A)))
        return np.ascontiguousarray(A)

    # ---- matrices --------------------------------------------------------------------------
    def set_matrix_data(self, M, Cm):
        M, Cm = _f64_colmajor(M), _f64_colmajor(Cm)
        if M.shape != Cm.shape or M.shape[0] != M.shape[1]:
            raise ValueError("M and C must be square and of equal size")
        self._check(self.L.clipper_hip_set_matrix(self.h, _dp(M), _dp(Cm), M.shape[0]))

    def set_sparse_matrix_data(self, m, Mcolptr, Mrow, Mval, Ccolptr, Crow, Cval):
        a = lambda x, t: np.ascontiguousarray(x, dtype=t)
        Mcp, Mr, Mv = a(Mcolptr, np.int64), a(Mrow, np.int32), a(Mval, np.float64)
        Ccp, Cr, Cv = a(Ccolptr, np.int64), a(Crow, np.int32), a(Cval, np.float64)
        self._check(self.L.clipper_hip_set_sparse(self.h, m, _i64p(Mcp), _ip(Mr), _dp(Mv),
                                                  _i64p(Ccp), _ip(Cr), _dp(Cv)))

    def get_affinity_matrix(self):
        m = self.m
        M = np.zeros((m, m), order="F")
        self._check(self.L.clipper_hip_get_matrix(self.h, _dp(M), None))
        return M

    def get_constraint_matrix(self):
        m = self.m
        Cm = np.zeros((m, m), order="F")
        self._check(self.L.clipper_hip_get_matrix(self.h, None, _dp(Cm)))
        return Cm

    def matvec(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64)
        yM, yC = np.zeros_like(x), np.zeros_like(x)
        self._check(self.L.clipper_hip_matvec(self.h, _dp(x), _dp(yM), _dp(yC)))
        return yM, yC

    # ---- solver ----------------------------------------------------------------------------
    def solve(self, u0):
        u0 = np.ascontiguousarray(u0, dtype=np.float64)
        n = self.m
        if n > 0 and u0.shape != (n,):
            raise ValueError(f"u0 must have shape ({n},)")
        u = np.zeros(max(n, 1))[:n]
        info = SolveInfo()
        self._check(self.L.clipper_hip_solve(self.h, _dp(u0), C.byref(self.params), _dp(u),
                                             C.byref(info)))
        nodes = np.zeros(max(info.num_nodes, 1), dtype=np.int32)
        k = self._check(self.L.clipper_hip_get_nodes(self.h, _ip(nodes), nodes.size))
        self.soln = Solution(t=info.seconds, ifinal=info.ifinal, nodes=nodes[:k].copy(), u0=u0,
                             u=u, score=info.score, d=info.d, n_passes=info.n_passes,
                             n_trials=info.n_trials)
        return self.soln

    def get_solution(self):
        return self.soln

    def get_selected_associations(self):
        k = len(self.soln.nodes)
        buf = np.zeros(2 * max(k, 1), dtype=np.int32)
        kk = self._check(self.L.clipper_hip_get_selected_associations(self.h, _ip(buf), max(k, 1)))
        if kk == 0:
            return np.zeros((0, 2), dtype=np.int32)
        return np.stack([buf[:kk], buf[kk:2 * kk]], axis=1)

    # ---- measurement ------------------------------------------------------------------------
    def densest_subgraph(self, S=None) -> np.ndarray:
        """dsd::solve(M_, S): exact densest subgraph of the current affinity matrix (all nodes, or
        restricted to the node list S)."""
        n = int(self.L.clipper_hip_num_associations(self.h))
        out = np.zeros(max(n, 1), dtype=np.int32)
        if S is None:
            k = self.L.clipper_hip_densest_subgraph(self.h, None, 0, _ip(out), len(out))
        else:
            Sa = np.ascontiguousarray(S, dtype=np.int32)
            k = self.L.clipper_hip_densest_subgraph(self.h, _ip(Sa), len(Sa), _ip(out), len(out))
        self._check(min(k, 0))
        return out[:k].copy()

    def set_window(self, window: int):
        """Line-search window (0 = automatic, 1 | 4 | 6 | 8); effective from the next build."""
        self._check(self.L.clipper_hip_set_window(self.h, int(window)))

    @property
    def window(self) -> int:
        return int(self.L.clipper_hip_window(self.h))

    def set_resident(self, mode: int):
        """0 = the resident (one-launch) solver where the slices fit on chip, 1 = never."""
        self._check(self.L.clipper_hip_set_resident(self.h, int(mode)))

    @property
    def last_solver(self) -> int:
        """What the last solve ran on: 0 = streaming launches, 1 = resident."""
        return int(self.L.clipper_hip_last_solver(self.h))

    def set_row_view(self, mode: int):
        """0 = build row views of M[live rows, :] during a solve where that pays, 1 = never."""
        self._check(self.L.clipper_hip_set_row_view(self.h, int(mode)))

    def set_subproblem(self, mode: int):
        """0 = hand a solve over to the live sub-problem (the associations that can still be selected) where that is
        provably exact and pays, 1 = never."""
        self._check(self.L.clipper_hip_set_subproblem(self.h, int(mode)))

    def view_matvec(self, rows, x):
        """(M_off[:, rows] x[rows], C_off[:, rows] x[rows]) through a row view built for `rows`."""
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=np.float64)
        yM, yC = np.zeros(self.m), np.zeros(self.m)
        self._check(self.L.clipper_hip_view_matvec(self.h, _ip(rows), rows.size, _dp(x), _dp(yM), _dp(yC)))
        return yM, yC

    def view_stats(self) -> ViewStats:
        t = ViewStats()
        self._check(self.L.clipper_hip_get_view_stats(self.h, C.byref(t)))
        return t

    @property
    def storage_in_use(self) -> int:
        return int(self.L.clipper_hip_storage_in_use(self.h))

    def set_profiling(self, on):
        """False / True, or 2: also HIP events around the launches of the resident solver on a view."""
        self._check(self.L.clipper_hip_set_profiling(self.h, int(on)))

    def timings(self) -> Timings:
        t = Timings()
        self._check(self.L.clipper_hip_get_timings(self.h, C.byref(t)))
        return t

    def bench_matvec(self, reps: int = 20) -> float:
        us = C.c_double()
        self._check(self.L.clipper_hip_bench_matvec(self.h, reps, C.byref(us)))
        return us.value

    def comm_init_callback(self, allgather):
        """exchange through `allgather(block: np.ndarray[float64]) -> np.ndarray` (the blocks of all
        ranks, rank order, concatenated) instead of RCCL — e.g. a torch.distributed gloo all-gather"""
        def thunk(user, send, recv, nbytes):
            try:
                n = nbytes // 8
                blk = np.ctypeslib.as_array(C.cast(send, C.POINTER(C.c_double)), shape=(n,))
                out = np.ascontiguousarray(allgather(blk.copy()), dtype=np.float64)
                np.ctypeslib.as_array(C.cast(recv, C.POINTER(C.c_double)), shape=(out.size,))[:] = out
                return 0
            except Exception:      # never unwind through the C frames
                import traceback
                traceback.print_exc()
                return 1
        self._xchg = ALLGATHER_FN(thunk)   # keep the trampoline alive
        self._check(self.L.clipper_hip_comm_init_callback(self.h, self._xchg, None))

    def debug_stamps(self):
        """per workgroup of the last pass launch: (start, decision done, end, info); needs
        CLIPPER_HIP_STAMPS=1 in the environment when the context is created"""
        rows = 16384 if os.environ.get("CLIPPER_HIP_STAMPS") == "2" else 4096
        out = np.zeros(rows * 4, dtype=np.int64)
        self._check(self.L.clipper_hip_debug_stamps(self.h, out.ctypes.data_as(C.POINTER(C.c_int64)), out.size))
        return out.reshape(rows, 4)

    def device_info(self):
        name = C.create_string_buffer(64)
        cus, hbm = C.c_int(), C.c_int64()
        self._check(self.L.clipper_hip_device_info(self.h, name, C.byref(cus), C.byref(hbm)))
        return name.value.decode(), cus.value, hbm.value


# ---- host-side neighbours of the path, through the C ABI (no device needed) ----------------------

def _status(rc):
    if rc < 0:
        raise ClipperError(f"clipper_hip error {rc}: {_last_error()}")
    return rc


def read_ply_xyz(path: str) -> np.ndarray:
    """utils::read_ply: vertex positions as a 3 x n float64 array (one datum per column)"""
    L = load_library()
    n = _status(L.clipper_hip_read_ply_xyz(path.encode(), None, 0))
    pts = np.zeros((n, 3), dtype=np.float64)
    _status(L.clipper_hip_read_ply_xyz(path.encode(), _dp(pts), n))
    return np.ascontiguousarray(pts.T)


def generate_synthetic_correspondences(n0: int, n1: int, Agood, m: int, rho: float, seed: int):
    """utils::generate_synthetic_correspondences -> (A m x 2, Agt ni x 2)"""
    L = load_library()
    Agood = np.asarray(Agood, dtype=np.int32).reshape(-1, 2)
    p = Agood.shape[0]
    gcm = np.ascontiguousarray(Agood.T).reshape(-1)
    A = np.zeros(2 * m, dtype=np.int32)
    Agt = np.zeros(2 * m, dtype=np.int32)
    ni = C.c_int64()
    _status(L.clipper_hip_generate_synthetic_correspondences(n0, n1, _ip(gcm), p, m, rho, seed, _ip(A), _ip(Agt),
                                                             C.byref(ni)))
    k = ni.value
    return A.reshape(2, m).T.copy(), Agt[:2 * k].reshape(2, k).T.copy()


def precision_recall(A, Agt):
    """utils::get_precision_recall"""
    L = load_library()
    A = np.asarray(A, dtype=np.int32).reshape(-1, 2)
    Agt = np.asarray(Agt, dtype=np.int32).reshape(-1, 2)
    a, g = np.ascontiguousarray(A.T).reshape(-1), np.ascontiguousarray(Agt.T).reshape(-1)
    p, r = C.c_double(), C.c_double()
    _status(L.clipper_hip_precision_recall(_ip(a) if a.size else None, A.shape[0], _ip(g) if g.size else None,
                                           Agt.shape[0], C.byref(p), C.byref(r)))
    return p.value, r.value


def estimate_rigid_transform(D1, D2, A) -> np.ndarray:
    """4 x 4 T with D2[:, A[i,1]] ~ R D1[:, A[i,0]] + t (D1, D2: 3 x n)"""
    L = load_library()
    D1, D2 = _f64_colmajor(D1), _f64_colmajor(D2)
    A = np.asarray(A, dtype=np.int32).reshape(-1, 2)
    a = np.ascontiguousarray(A.T).reshape(-1)
    T = np.zeros(16, dtype=np.float64)
    _status(L.clipper_hip_estimate_rigid_transform(_dp(D1), D1.shape[1], _dp(D2), D2.shape[1], _ip(a), A.shape[0],
                                                   _dp(T)))
    return T.reshape(4, 4).T.copy()


def device_count() -> int:
    return int(load_library().clipper_hip_device_count())


def debug_occupy(device: int, workgroups: int, lds_bytes: int, milliseconds: float) -> None:
    """Test infrastructure: `workgroups` wave slots with `lds_bytes` of LDS each are held for `milliseconds`
    (blocks until the kernel is over): another tenant on the device."""
    rc = load_library().clipper_hip_debug_occupy(device, workgroups, lds_bytes, float(milliseconds))
    if rc < 0:
        raise ClipperError(f"clipper_hip_debug_occupy: {rc} ({_last_error()})")


def _last_error() -> str:
    L = load_library()
    return (L.clipper_hip_last_error() or b"").decode()


def knn(P0, P1, knn: int, device: int = 0):
    """k nearest neighbours in P1 (d x n1) of every point of P0 (d x n0), columns = points as
    `clipper::Data`. Returns (idx n0 x knn int32, sqd n0 x knn)."""
    L = load_library()
    P0c, P1c = _f64_colmajor(P0), _f64_colmajor(P1)
    d, n0 = P0c.shape
    n1 = P1c.shape[1]
    idx = np.zeros((n0, knn), dtype=np.int32)
    sqd = np.zeros((n0, knn), dtype=np.float64)
    rc = L.clipper_hip_knn(device, _dp(P0c), n0, _dp(P1c), n1, d, knn, _ip(idx), _dp(sqd))
    if rc != 0:
        raise RuntimeError(f"clipper_hip error {rc}: {_last_error()}")
    return idx, sqd


def distance_based_correspondences(P0, P1, knn: int, radius: float, enforce_1to1: bool,
                                   device: int = 0) -> np.ndarray:
    """utils::distance_based_correspondences of the reference benchmark (bm_utils.cpp:147-232) on
    the device. P0: d x n0, P1: d x n1 (columns = points). Returns the associations n x 2."""
    L = load_library()
    P0c, P1c = _f64_colmajor(P0), _f64_colmajor(P1)
    d, n0 = P0c.shape
    n1 = P1c.shape[1]
    cap = n0 * knn
    buf = np.zeros(2 * max(cap, 1), dtype=np.int32)
    n = L.clipper_hip_distance_based_correspondences(device, _dp(P0c), n0, _dp(P1c), n1, d, knn,
                                                     float(radius), int(bool(enforce_1to1)),
                                                     _ip(buf), cap)
    if n < 0:
        raise RuntimeError(f"clipper_hip error {n}: {_last_error()}")
    n = int(n)
    return np.stack([buf[:n], buf[n:2 * n]], axis=1).astype(np.int32)
