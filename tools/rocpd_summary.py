#!/usr/bin/env python3
"""Summarise a rocprofv3 (ROCm 7.2 rocpd SQLite) result: per-kernel calls / total / avg / min / max / share,
plus registers, LDS and grid.  usage: rocpd_summary.py <results.db> [--pmc]"""
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name.replace("void figh::", "").replace("figh::", "")


def main():
    db = sys.argv[1]
    con = sqlite3.connect(db)
    cur = con.cursor()
    rows = cur.execute("select name, duration, grid_x, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, "
                       "scratch_size from kernels").fetchall()
    agg = {}
    for name, dur, gx, wx, lds, vg, ag, sg, scr in rows:
        # one line per (kernel, launch shape): the full-size TSQR launch and the one-wave regrouped factorisation share a
        # kernel name, and so do the merge levels
        key = "%s [grid %d]" % (short(name), gx)
        a = agg.setdefault(key, {"n": 0, "tot": 0, "min": 1 << 62, "max": 0, "meta": (gx, wx, lds, vg, ag, sg, scr)})
        a["n"] += 1
        a["tot"] += dur
        a["min"] = min(a["min"], dur)
        a["max"] = max(a["max"], dur)
    total = sum(a["tot"] for a in agg.values()) or 1
    print("# rocprofv3 --kernel-trace --stats summary of %s" % db)
    print("%-66s %6s %12s %11s %11s %11s %6s  %s" % ("kernel [launch shape]", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "grid/wg lds vgpr agpr sgpr scratch"))
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1]["tot"]):
        print("%-66s %6d %12.1f %11.2f %11.2f %11.2f %6.2f  %s" % (
            k[:66], a["n"], a["tot"] / 1e3, a["tot"] / a["n"] / 1e3, a["min"] / 1e3, a["max"] / 1e3, 100.0 * a["tot"] / total,
            "%d/%d %d %d %d %d %d" % a["meta"]))
    if "--pmc" in sys.argv:
        try:
            q = cur.execute("select k.name, p.counter_name, avg(p.value), count(*) from pmc_events p join kernels k on "
                            "p.dispatch_id = k.dispatch_id group by k.name, p.counter_name").fetchall()
            print("\n# PMC (average per dispatch)")
            for name, cname, val, n in q:
                print("%-52s %-28s %18.1f  (%d dispatches)" % (short(name)[:52], cname, val, n))
        except Exception as e:  # schema differences between ROCm releases
            print("# PMC query failed:", e)


if __name__ == "__main__":
    main()
