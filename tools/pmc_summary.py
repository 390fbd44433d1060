#!/usr/bin/env python3
"""Per-kernel PMC summary from separate rocprofv3 --pmc passes (rocpd SQLite).  For each kernel the dispatch with
the largest duration is reported (the full-size launch).  FETCH_SIZE is doubled, as MI355X_MICROARCH.md
prescribes for gfx950 (128-B requests tallied at 64 B); FETCH_SIZE / WRITE_SIZE are in KiB.
usage: pmc_summary.py out.json db1 [db2 ...]"""
import json
import re
import sqlite3
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name).replace("void figh::", "").replace("figh::", "")


def main():
    out, dbs = sys.argv[1], sys.argv[2:]
    res = {}
    for db in dbs:
        cur = sqlite3.connect(db).cursor()
        rows = cur.execute("select name, counter_name, dispatch_id, sum(counter_value), max(duration) from pmc_events "
                           "group by name, counter_name, dispatch_id").fetchall()
        for name, cname, did, val, dur in rows:
            k = res.setdefault(short(name), {})
            if cname not in k or dur > k[cname][1]:
                k[cname] = (val, dur)
    summary = {}
    for kern, ctrs in res.items():
        d = {c: v for c, (v, _) in ctrs.items()}
        d["duration_us_in_pmc_pass"] = max(dur for _, dur in ctrs.values()) / 1e3
        if "FETCH_SIZE" in d:
            d["hbm_read_bytes"] = 2.0 * d["FETCH_SIZE"] * 1024.0   # gfx950 correction: x2
        if "WRITE_SIZE" in d:
            d["hbm_write_bytes"] = d["WRITE_SIZE"] * 1024.0
        if "hbm_read_bytes" in d and "hbm_write_bytes" in d:
            d["hbm_bytes"] = d["hbm_read_bytes"] + d["hbm_write_bytes"]
        summary[kern] = d
    with open(out, "w") as f:
        json.dump(summary, f, indent=1, sort_keys=True)
    for kern in sorted(summary, key=lambda k: -summary[k]["duration_us_in_pmc_pass"])[:6]:
        print(kern, json.dumps({k: (round(v, 1) if isinstance(v, float) else v) for k, v in summary[kern].items()}))


if __name__ == "__main__":
    main()
