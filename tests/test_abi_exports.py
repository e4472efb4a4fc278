"""CPU-side checks of the drop-in boundary: the product library loads and exports every
symbol include/clipper_hip.h declares (no compute calls — there is no GPU here)."""
import ctypes
import os
import re

from clipper_amd import _abi as abi
from clipper_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "clipper_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(clipper_hip_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert _declared_functions() == sorted(abi.EXPORTED_SYMBOLS)


def test_library_exports_every_declared_symbol():
    path = build.build_hip()          # cross-compiles for gfx950 if stale; no GPU needed
    lib = ctypes.CDLL(path)
    for name in _declared_functions():
        assert hasattr(lib, name), f"{name} is declared in clipper_hip.h but not exported"


def test_struct_layouts_match_header():
    # clipper_params_t: 3 doubles, 2 int32, double, int32 (+pad), 2 doubles, 2 int32
    # (sizes / offsets printed by a C program including include/clipper_hip.h)
    assert ctypes.sizeof(abi.Params) == 72
    assert abi.Params.rounding.offset == 68 and abi.Params.beta.offset == 32
    assert ctypes.sizeof(abi.SolveInfo) == 48
    assert ctypes.sizeof(abi.Timings) == 56
    p = abi.Params()
    assert (p.tol_u, p.tol_F, p.maxiniters, p.maxoliters, p.beta, p.maxlsiters, p.eps,
            p.affinityeps, p.rescale_u0, p.rounding) == (
        1e-8, 1e-9, 200, 1000, 0.25, 99, 1e-9, 1e-4, 1, 2)   # clipper.h:27-60


def test_product_does_not_reference_the_oracle():
    # the product path must never import, link or call anything under oracle/
    bad = []
    for dp, _, fs in os.walk(os.path.join(ROOT, "clipper_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"clipper_ref|from oracle|import oracle|oracle/", t):
                    bad.append(os.path.join(dp, f))
    assert not bad, f"product files mention the oracle: {bad}"
