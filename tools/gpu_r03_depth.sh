#!/bin/bash
# steps in flight / occupancy of the pass kernel, for passes on M and on a row view (library variants under tools/_bin,
# built with -DCLIPPER_SL_D=.. -DCLIPPER_SL_OCC=..): tools/rowview_probe.py --profile at m = 10k and 100k
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for name in product d4o6 d4o5 d6o4 d8o4; do
  lib=tools/_bin/libclipper_hip_$name.so; [ $name = product ] && lib=clipper_amd/lib/libclipper_hip.so
  CLIPPER_HIP_LIB=$PWD/$lib timeout 100 python tools/rowview_probe.py --m 10000 100000 --reps 3 --profile > gpurun_out/r03k_depth_$name.jsonl 2> gpurun_out/r03k_depth_$name.err
  python - $name gpurun_out/r03k_depth_$name.jsonl <<'PY'
import json, sys
for l in open(sys.argv[2]):
    try: r = json.loads(l)
    except Exception: continue
    on, off = r["on"], r["off"]
    print(sys.argv[1], r["m"], "views on: solve", on["solve_ms"], "pass_us", on["pass_us"], "view_pass_us", on["view_pass_us"], "| views off: solve", off["solve_ms"], "pass_us", off["pass_us"], "nodes", on["nodes_sha"])
PY
done
