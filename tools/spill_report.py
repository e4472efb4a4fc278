#!/usr/bin/env python3
"""Where a kernel of clipper_hip.hip spills: compiles the device code to assembly with line tables and
lists, for the kernels whose mangled name contains the given substrings, every scratch load / store by
source line, marking the ones that sit inside a loop that contains fp64 fmas (the streaming loops).
  python tools/spill_report.py k_gemv_slicesIfLi1ELi6 [more substrings]"""
import os
import re
import subprocess
import sys
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = "/tmp/clipper_dev.s"
subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
                "-gline-tables-only", "-S", "--cuda-device-only", "-w", "-o", out,
                os.path.join(ROOT, "clipper_amd/csrc/clipper_hip.hip")] + sys.argv[2:] * 0, check=True)
s = open(out).read()
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', s):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split("/")[-1]
for m in re.finditer(r"^(_ZN11clipper_hip\w+):", s, flags=re.M):
    name = m.group(1)
    if not any(sub in name for sub in sys.argv[1:]):
        continue
    i = m.start()
    j = s.index(".Lfunc_end", i)
    body = s[i:j].split("\n")
    labels = {}
    for k, l in enumerate(body):
        mm = re.match(r"^(\.LBB\d+_\d+):", l)
        if mm:
            labels[mm.group(1)] = k
    loops = []
    for k, l in enumerate(body):
        mm = re.search(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", l)
        if mm:
            t = mm.group(1) or mm.group(2)
            if t in labels and labels[t] < k:
                loops.append((labels[t], k))
    hot = [(a, b) for a, b in loops if sum("v_fma_f64" in body[t] for t in range(a, b + 1)) >= 7]
    cur = None
    c = Counter()
    for k, l in enumerate(body):
        mm = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if mm:
            cur = (files.get(int(mm.group(1)), "?"), int(mm.group(2)))
        if "scratch_" in l:
            inner = [(a, b) for a, b in hot if a <= k <= b]
            depth = "HOT" if inner and min(b - a for a, b in inner) < 400 else ("loop" if inner else "")
            c[(cur, "store" if "store" in l else "load", depth)] += 1
    print(name, ": scratch ops", sum(c.values()), " inside small fma loops:", sum(v for (a, b, d), v in c.items() if d == "HOT"))
    for k, v in sorted(c.items(), key=lambda t: (str(t[0][0]), t[0][1])):
        print("   ", k, v)
