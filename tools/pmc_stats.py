#!/usr/bin/env python3
"""Sums a rocprofv3 --pmc counter per mifx kernel from the results database: python tools/pmc_stats.py <dir> <COUNTER> [--schema]"""
import glob
import os
import sqlite3
import sys


def main():
    path, counter = sys.argv[1], sys.argv[2]
    db = sqlite3.connect(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)[0])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    if "--schema" in sys.argv:
        for t in tabs:
            if t in ("counters_collection",):
                cols = [d[0] for d in cur.execute(f"select * from {t} limit 1").description]
                print(t, cols)
        return
    # rocpd view `counters_collection`: one row per (dispatch, counter)
    rows = cur.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection where counter_name = ? group by kernel_name", (counter,)).fetchall()
    print(f"{'kernel':64s} {'dispatches':>10s} {counter + ' (sum)':>22s} {'per dispatch':>16s}")
    for name, cname, total, n in sorted(rows, key=lambda r: -r[2]):
        if "mifx::" in name and "ibl_" not in name:
            short = name.split("(")[0].replace("void ", "")
            print(f"{short:64s} {n:10d} {total:22.1f} {total / n:16.1f}")


if __name__ == "__main__":
    main()
