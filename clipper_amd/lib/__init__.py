"""Built native artefacts live here (git-ignored, shipped to the GPU box by gpurun):
libclipper_hip.so (HIP kernels + C ABI) and clipperpy*.so (pybind11 module)."""
