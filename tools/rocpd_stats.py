#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table,
the same content `--stats` prints for CSV output: calls, total, mean, min, max, share.
Usage: tools/rocpd_stats.py <results.db> [--min-us X] [--json out.json]
Kernels whose duration is below --min-us (default 0) can be reported separately: the solver
deliberately launches no-op kernels after convergence (they exit on the `done` flag)."""
import json
import sqlite3
import sys


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    rows = con.execute("select name, duration, grid_x, grid_y, workgroup_x from kernels").fetchall()
    agg = {}
    for name, dur, gx, gy, wx in rows:
        short = name.split("(")[0]
        a = agg.setdefault(short, [])
        a.append(dur)
    total = sum(sum(v) for v in agg.values())
    out = []
    print(f"{'kernel':70s} {'calls':>7s} {'total_ms':>10s} {'mean_us':>9s} {'median_us':>9s} {'min_us':>8s} {'max_us':>8s} {'share':>6s}")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        v = sorted(v)
        rec = dict(kernel=k, calls=len(v), total_ms=sum(v) / 1e6, mean_us=sum(v) / len(v) / 1e3,
                   median_us=v[len(v) // 2] / 1e3, min_us=v[0] / 1e3, max_us=v[-1] / 1e3,
                   share=sum(v) / total)
        out.append(rec)
        print(f"{k[:70]:70s} {rec['calls']:7d} {rec['total_ms']:10.3f} {rec['mean_us']:9.2f} "
              f"{rec['median_us']:9.2f} {rec['min_us']:8.2f} {rec['max_us']:8.2f} {100 * rec['share']:5.1f}%")
    # the solver's kernels are also launched for transition iterations (one workgroup works)
    # and, a few times per solve, past convergence (immediate exit): the PASS launches — the
    # ones bench.py's HIP events time — are those of at least half the median duration
    print()
    print("pass launches only (duration >= 0.5 x median):")
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if not ("k_gemv" in k or "k_pass" in k or "k_tail" in k):
            continue
        v = sorted(v)
        if "k_gemv" in k or "k_pass" in k:
            # three populations since round 3: passes on M, passes on a row view of M (far fewer bytes:
            # shorter), and launches that do nothing (a few us). On M: at least 0.6 x the longest.
            groups = (("passes on M", [x for x in v if x >= 0.6 * v[-1]]),
                      ("shorter launches (passes on a row view, transitions)", [x for x in v if 4000 <= x < 0.6 * v[-1]]))
        else:
            med = v[len(v) // 2]
            groups = (("pass launches", [x for x in v if x >= 0.5 * med]),)
        for label, w in groups:
            if not w:
                continue
            print(f"{(k[:48] + ' [' + label + ']')[:100]:100s} {len(w):6d} launches  mean {sum(w) / len(w) / 1e3:9.2f} us  "
                  f"median {w[len(w) // 2] / 1e3:9.2f} us  min {w[0] / 1e3:8.2f}  max {w[-1] / 1e3:8.2f}")
            out.append(dict(kernel=k + " [" + label + "]", calls=len(w), mean_us=sum(w) / len(w) / 1e3,
                            median_us=w[len(w) // 2] / 1e3, min_us=w[0] / 1e3, max_us=w[-1] / 1e3))
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
