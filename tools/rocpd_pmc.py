#!/usr/bin/env python3
"""Per-kernel summary of a PMC counter (FETCH_SIZE / WRITE_SIZE, in KB) from rocprofv3 rocpd
databases. Launches shorter than 3 us (the solver's post-convergence no-ops) are excluded.
Usage: tools/rocpd_pmc.py <results.db> [...]"""
import sqlite3
import sys
from collections import defaultdict

for db in sys.argv[1:]:
    con = sqlite3.connect(db)
    rows = con.execute(
        "select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    agg = defaultdict(list)
    for name, cname, val, dur in rows:
        short = name.split("(")[0].replace("void clipper_hip::", "").replace("clipper_hip::", "")
        if dur < 3000 and short.startswith(("k_gemv", "k_pass", "k_tail", "k_reduce")):
            continue
        agg[(short, cname)].append(val)
    print(db)
    for (k, c), v in sorted(agg.items()):
        v = sorted(v)
        print(f"  {k[:44]:44s} {c:11s} n={len(v):4d} median {v[len(v)//2]:12.1f} KB  "
              f"mean {sum(v)/len(v):12.1f} KB  min {v[0]:12.1f}  max {v[-1]:12.1f}")
