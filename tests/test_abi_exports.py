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
    assert ctypes.sizeof(abi.Timings) == 96
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


def test_enumerators_match_header(tmp_path):
    """the storage / rounding / error enumerators of the C headers against the ctypes constants
    and the C++ facade's Storage enum (a C program prints them; gcc only, no GPU)."""
    import subprocess
    src = tmp_path / "enums.c"
    src.write_text('#include <stdio.h>\n#include "clipper_hip.h"\nint main(void){printf("%d %d %d %d %d %d %d %d %d\\n",'
                   'CLIPPER_HIP_STORE_F32, CLIPPER_HIP_STORE_F64, CLIPPER_HIP_STORE_F32_CSC, CLIPPER_HIP_STORE_F64_CSC,'
                   'CLIPPER_ROUNDING_NONZERO, CLIPPER_ROUNDING_DSD, CLIPPER_ROUNDING_DSD_HEU,'
                   'CLIPPER_HIP_E_NOMEM, CLIPPER_HIP_E_SCOPE);return 0;}\n')
    exe = tmp_path / "enums"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    assert got[:4] == [abi.STORE_F32, abi.STORE_F64, abi.STORE_F32_CSC, abi.STORE_F64_CSC] == [0, 1, 2, 3]
    assert got[4:7] == [abi.ROUNDING_NONZERO, abi.ROUNDING_DSD, abi.ROUNDING_DSD_HEU]
    assert got[7:] == [-2, -7]
    facade = open(os.path.join(ROOT, "include", "clipper", "clipper.h")).read()
    assert "enum class Storage { F32 = 0, F64 = 1, F32_CSC = 2, F64_CSC = 3 }" in facade


def test_every_entry_point_is_guarded():
    """the C ABI must not throw. Every function include/clipper_hip.h declares is defined as a
    function-try-block closed by a CLIPPER_HIP_GUARD_* macro (csrc/host_state.hpp: std::bad_alloc ->
    CLIPPER_HIP_E_NOMEM, anything else -> CLIPPER_HIP_E_INTERNAL). Source scan: pairing and return kind."""
    csrc = os.path.join(ROOT, "clipper_amd", "csrc")
    text = "\n".join(open(os.path.join(csrc, f)).read() for f in ("clipper_hip.hip", "host_registration.hpp"))
    kinds = {"int": "INT", "int64_t": "INT", "clipper_hip_t*": "PTR", "const char*": "STR", "void": "VOID"}
    for name in _declared_functions():
        m = re.search(r"^(const char\*|clipper_hip_t\*|void|int64_t|int)\s+" + name + r"\([^;{]*\)\s*try\s*\{", text, flags=re.M)
        assert m, f"{name}: no definition of the form `T {name}(...) try {{`"
        end = re.search(r"^\}(.*)$", text[m.end():], flags=re.M)       # the function's closing brace (column 0)
        assert end and end.group(1).strip() == "CLIPPER_HIP_GUARD_" + kinds[m.group(1)], \
            f"{name}: closed by {end.group(1).strip() if end else None!r}"
    guard = open(os.path.join(csrc, "host_state.hpp")).read()
    assert "catch (const std::bad_alloc&)" in guard and "catch (...)" in guard
    assert "CLIPPER_HIP_E_INTERNAL" in open(os.path.join(ROOT, "include", "clipper_hip.h")).read()


def test_env_knobs_are_the_documented_ones():
    """Every CLIPPER_HIP_* environment variable the product reads is listed in INTEGRATION.md ("Runtime notes"), and
    nothing is listed that is not read (harness knobs do not live in the product)."""
    read = set()
    for dp, _, fs in os.walk(os.path.join(ROOT, "clipper_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp", ".hpp")):
                t = open(os.path.join(dp, f), errors="ignore").read()
                read |= set(re.findall(r'getenv\(\s*"(CLIPPER_HIP_[A-Z0-9_]+)"', t))
                read |= set(re.findall(r'environ(?:\.get)?[\[(]\s*"(CLIPPER_HIP_[A-Z0-9_]+)"', t))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("Environment knobs"):]
    listed = set(re.findall(r"`(CLIPPER_HIP_[A-Z0-9_]+)`", table))
    assert read == listed, (sorted(read - listed), sorted(listed - read))
