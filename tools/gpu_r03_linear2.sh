#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for name in product lin_o6 lin_o5; do
  lib=tools/_bin/libclipper_hip_$name.so; [ $name = product ] && lib=clipper_amd/lib/libclipper_hip.so
  CLIPPER_HIP_LIB=$PWD/$lib timeout 60 python tools/pass_min_probe.py 10000 100000 2>/dev/null | tee -a gpurun_out/r03m_pass_min.jsonl
done
