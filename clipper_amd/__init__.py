"""clipper_amd — MI355X-native implementation of CLIPPER's dense-cluster hot path
(affinity build + projected-gradient solver). See DESIGN.md.

  clipper_amd/csrc/        hand-written gfx950 HIP kernels + the C ABI (include/clipper_hip.h)
  clipper_amd/csrc/host/   C++ facade `clipper::CLIPPER` + pybind11 module `clipperpy`
  clipper_amd/_abi.py      ctypes view of the C ABI (tests, bench)
  clipper_amd/synth.py     seeded synthetic registration problems (measurement recipe)
"""
__version__ = "0.1.0"


def load_clipperpy():
    """Import the pybind11 module `clipperpy` (the reference's Python surface) from
    clipper_amd/lib and register it under its reference name, so `import clipperpy` works."""
    import importlib
    import sys

    mod = importlib.import_module("clipper_amd.lib.clipperpy")
    sys.modules.setdefault("clipperpy", mod)
    for sub in ("invariants", "utils", "dsd"):
        sys.modules.setdefault(f"clipperpy.{sub}", getattr(mod, sub))
    return mod
