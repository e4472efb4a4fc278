#!/usr/bin/env python3
"""One affinity+solve step as the GPU saw it: every kernel and copy of the last step of a
`rocprofv3 --kernel-trace --memory-copy-trace` run of bench.py, with start offsets and gaps.
  python tools/step_timeline.py <results.db> [index of the step's fill launch]"""
import sqlite3
import sys


def main():
    con = sqlite3.connect(sys.argv[1])
    ev = [(s, e, n.split("(")[0].replace("void clipper_hip::", "").replace("clipper_hip::", ""))
          for n, s, e in con.execute("select name, start, end from kernels")]
    try:
        ev += [(s, e, "copy " + str(n)) for n, s, e in con.execute("select name, start, end from memory_copies")]
    except sqlite3.Error as ex:
        print("no memory_copies view:", ex)
    ev.sort()
    fills = [i for i, x in enumerate(ev) if x[2].startswith("k_affinity_sym")]
    if len(fills) < 2:
        print("need two steps")
        return
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(fills) - 2   # which step (index of its fill)
    i0, i1 = fills[k], fills[k + 1]
    while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 30000 and not ev[i0 - 1][2].startswith("k_tail") and not ev[i0 - 1][2].startswith("k_gemv"):
        i0 -= 1
    t0 = ev[i0][0]
    prev_end = t0
    rows = ev[i0:i1]
    busy = 0
    print(f"{'start us':>9s} {'dur us':>8s} {'gap us':>7s}  event")
    shown = 0
    for k, (s, e, n) in enumerate(rows):
        gap = (s - prev_end) / 1e3
        busy += (e - s)
        if shown < 14 or k >= len(rows) - 6 or gap > 3.0:
            print(f"{(s - t0) / 1e3:9.2f} {(e - s) / 1e3:8.2f} {gap:7.2f}  {n[:60]}")
            shown += 1
        prev_end = max(prev_end, e)
    span = (rows[-1][1] - t0) / 1e3
    print(f"span {span:.1f} us, busy {busy / 1e3:.1f} us, idle {span - busy / 1e3:.1f} us, events {len(rows)}")


if __name__ == "__main__":
    main()
