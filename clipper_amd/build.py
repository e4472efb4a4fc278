"""Build recipe for the native parts (explicit hipcc / g++ commands, in-tree outputs).

  clipper_amd/lib/libclipper_hip.so   HIP kernels + C ABI, gfx950 only
  clipper_amd/lib/clipperpy*.so       pybind11 module over the C++ facade (if sources exist)

The built .so files are git-ignored but travel to the GPU box with the gpurun snapshot.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
HIP_LIB = os.path.join(LIBDIR, "libclipper_hip.so")

HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
# -ffp-contract=off: fp64 expressions round as written, fma only where spelled out (shared
# convention with the oracle, see kernels.hip.h)
HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
             "-shared", "-Wall", "-Wno-unused-value"]


def _newer(target: str, sources: list[str]) -> bool:
    if not os.path.exists(target):
        return False
    t = os.path.getmtime(target)
    return all(os.path.getmtime(s) <= t for s in sources if os.path.exists(s))


def _run(cmd: list[str]):
    print("+", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)


def build_hip(force: bool = False) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    srcs = [os.path.join(CSRC, "clipper_hip.hip"), os.path.join(CSRC, "kernels.hip.h"),
            *[os.path.join(CSRC, f) for f in ("k_solver.hip.h", "k_gemv.hip.h", "k_csc.hip.h", "k_slices.hip.h", "k_resident.hip.h", "host_resident.hpp", "k_rv_resident.hip.h", "host_rv_resident.hpp",
                                              "k_affinity.hip.h", "k_matrix.hip.h", "k_rowview.hip.h", "k_subproblem.hip.h", "k_knn.hip.h")],
            *[os.path.join(CSRC, f) for f in ("host_state.hpp", "host_solver.hpp", "host_matrix.hpp", "host_plan.hpp", "host_batch.hpp", "host_rowview.hpp", "host_subproblem.hpp", "host_registration.hpp")],
            os.path.join(CSRC, "dsd_host.h"),
            os.path.join(ROOT, "include", "clipper_hip.h"),
            os.path.join(ROOT, "include", "clipper_abi.h")]
    if force or not _newer(HIP_LIB, srcs):
        _run([HIPCC, *HIP_FLAGS, "-o", HIP_LIB, srcs[0], "-ldl"])
    return HIP_LIB


def pymodule_path() -> str:
    return os.path.join(LIBDIR, "clipperpy" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))


def build_pymodule(force: bool = False) -> str | None:
    """pybind11 module `clipperpy` (reference: bindings/python/py_clipper.cpp) over the C++
    facade in csrc/host/. Links against libclipper_hip.so through an $ORIGIN rpath."""
    host = os.path.join(CSRC, "host")
    src = os.path.join(host, "py_clipper.cpp")
    if not os.path.exists(src):
        return None
    import pybind11
    out = pymodule_path()
    srcs = [src, os.path.join(host, "clipper.cpp")] + [
        os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(ROOT, "include")) for f in fs]
    if force or not _newer(out, srcs + [HIP_LIB]):
        _run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-fopenmp",
              # the module's type casters are written for the stand-in containers of
              # include/clipper/types.h: choose that type family explicitly
              "-DCLIPPER_NO_EIGEN",
              "-I", os.path.join(ROOT, "include"), "-I", pybind11.get_include(),
              "-I", sysconfig.get_paths()["include"],
              src, os.path.join(host, "clipper.cpp"),
              "-L", LIBDIR, "-lclipper_hip", "-Wl,-rpath,$ORIGIN", "-o", out])
    return out


def build_all(force: bool = False):
    build_hip(force)
    build_pymodule(force)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
