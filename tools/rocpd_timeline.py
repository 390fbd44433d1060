#!/usr/bin/env python3
"""Timeline of the LAST step in a rocprofv3 (rocpd SQLite) kernel trace: every dispatch from the last launch of the kernel whose
name contains <marker> on -- start offset, duration, idle gap in front of it (all in microseconds) -- and the sums.
usage: rocpd_timeline.py <results.db> <marker>"""
import sqlite3
import sys

from rocpd_summary import short


def main():
    db, marker = sys.argv[1], sys.argv[2]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, start, end, grid_x from kernels order by start").fetchall()
    starts = [i for i, r in enumerate(rows) if marker in r[0]]
    if len(starts) < 2:
        raise SystemExit("fewer than two launches of a kernel matching %r" % marker)
    seg = rows[starts[-2]:starts[-1]]  # the last complete step
    t0, prev_end, busy = seg[0][1], seg[0][1], 0
    print("# %s: step of %d dispatches, %.1f us from its first launch to the next step's" % (db, len(seg), (rows[starts[-1]][1] - t0) / 1e3))
    print("%9s %9s %8s  %s" % ("start", "dur", "gap", "kernel [grid]"))
    for name, s, e, gx in seg:
        print("%9.1f %9.1f %8.1f  %s [%d]" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, short(name)[:70], gx))
        busy += e - s
        prev_end = max(prev_end, e)
    print("# busy %.1f us, idle between dispatches %.1f us, tail to the next step %.1f us" % (
        busy / 1e3, (prev_end - t0 - busy) / 1e3, (rows[starts[-1]][1] - prev_end) / 1e3))


if __name__ == "__main__":
    main()
