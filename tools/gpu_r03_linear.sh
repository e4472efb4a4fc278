#!/bin/bash
# the LINEAR WINDOW (k_slices.hip.h, CLIPPER_SL_XMODE=3) as library variants against the product: pass times, then the
# parity tests through the variant (CLIPPER_HIP_LIB)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for name in product lin_o6 lin_o5; do
  lib=tools/_bin/libclipper_hip_$name.so; [ $name = product ] && lib=clipper_amd/lib/libclipper_hip.so
  CLIPPER_HIP_LIB=$PWD/$lib timeout 100 python tools/rowview_probe.py --m 10000 100000 --reps 3 --profile > gpurun_out/r03l_lin_$name.jsonl 2> gpurun_out/r03l_lin_$name.err
  python - $name gpurun_out/r03l_lin_$name.jsonl <<'PY'
import json, sys
for l in open(sys.argv[2]):
    try: r = json.loads(l)
    except Exception: continue
    on, off = r["on"], r["off"]
    print(sys.argv[1], r["m"], "views on: solve", on["solve_ms"], "pass_us", on["pass_us"], "view_pass_us", on["view_pass_us"], "trials", on["trials"], "| views off: solve", off["solve_ms"], "pass_us", off["pass_us"], "trials", off["trials"], "score", on["score"], "nodes", on["nodes_sha"])
PY
done
for name in lin_o6; do
  CLIPPER_HIP_LIB=$PWD/tools/_bin/libclipper_hip_$name.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_rowview.py tests/test_gpu_fuzz.py tests/test_gpu_configs.py -q -k "not cfg5 and not clipperpy" > gpurun_out/r03l_tests_$name.txt 2>&1
  echo "tests through $name: rc $?"; tail -15 gpurun_out/r03l_tests_$name.txt | cut -c1-250
done
