#!/usr/bin/env python3
"""Timeline of the long kernels in a rocprofv3 --kernel-trace results .db: start / end of every dispatch
longer than MIN_US relative to the first one, and for each how long it ran beside another long kernel.
  python tools/prof_timeline.py x_results.db [min_us] [max_rows]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    min_us = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
    max_rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    s_col = "start" if "start" in cols else "start_time"
    e_col = "end" if "end" in cols else "end_time"
    rows = cur.execute("select name, %s, %s, duration from kernels where duration > ? order by %s" % (s_col, e_col, s_col),
                       (min_us * 1e3,)).fetchall()
    if not rows:
        print("no kernels; columns:", cols)
        return
    t0 = rows[0][1]
    rows = rows[-max_rows:]
    print("%-46s %10s %10s %9s %9s" % ("kernel", "start_us", "end_us", "dur_us", "beside_us"))
    for i, (name, s, e, d) in enumerate(rows):
        beside = 0
        for j, (n2, s2, e2, d2) in enumerate(rows):
            if i != j:
                beside += max(0, min(e, e2) - max(s, s2))
        print("%-46s %10.1f %10.1f %9.1f %9.1f" % (name.split("(")[0].replace("void mi::", "")[:46], (s - t0) / 1e3, (e - t0) / 1e3, d / 1e3, beside / 1e3))


if __name__ == "__main__":
    main()
