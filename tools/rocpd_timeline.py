#!/usr/bin/env python3
"""Per-iteration timeline of the solve loop from a rocprofv3 rocpd database: for the real
(non no-op) iterations prints mean kernel durations and the gaps between consecutive kernels.
Usage: tools/rocpd_timeline.py <results.db>"""
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
rows = con.execute("select name, start, end from kernels order by start").fetchall()
seq = [(n.split("(")[0].replace("void clipper_hip::", "").replace("clipper_hip::", ""), s, e)
       for n, s, e in rows]
dur = defaultdict(list)
gap = defaultdict(list)
for k in range(1, len(seq)):
    n0, s0, e0 = seq[k - 1]
    n1, s1, e1 = seq[k]
    if (e1 - s1) < 3000 and n1.startswith("k_gemv"):
        continue  # no-op tail
    if n1.startswith(("k_gemv", "k_pass", "k_reduce", "k_tail")) and n0.startswith(("k_gemv", "k_pass", "k_reduce", "k_tail")):
        if (e0 - s0) < 2500 and n0.startswith("k_gemv"):
            continue
        gap[f"{n0[:8]} -> {n1[:8]}"].append(s1 - e0)
for n, s, e in seq:
    if n.startswith("k_gemv") and (e - s) < 3000:
        continue
    dur[n].append(e - s)
for n, v in dur.items():
    v = sorted(v)
    print(f"dur  {n[:40]:40s} n={len(v):5d} median {v[len(v)//2]/1e3:8.2f} us  mean {sum(v)/len(v)/1e3:8.2f}")
for n, v in gap.items():
    v = sorted(v)
    print(f"gap  {n:40s} n={len(v):5d} median {v[len(v)//2]/1e3:8.2f} us  mean {sum(v)/len(v)/1e3:8.2f}")
