#!/usr/bin/env python3
"""Per-kernel rocprofv3 --pmc counters from the results database, averaged per dispatch:  python tools/pmc_stats.py <dir> COUNTER [COUNTER ...]"""
import collections
import glob
import os
import sqlite3
import sys


def main():
    path, counters = sys.argv[1], sys.argv[2:]
    db = sqlite3.connect(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)[0])
    # rocpd view `counters_collection`: one row per (dispatch, counter); columns kernel_name, counter_name, value, duration ...
    rows = db.execute("select kernel_name, counter_name, sum(value), count(*), sum(duration) from counters_collection group by kernel_name, counter_name").fetchall()
    tab = collections.defaultdict(dict)
    for name, cname, total, n, dur in rows:
        if "mifx::" in name and "ibl_" not in name:
            short = name.split("(")[0].replace("void ", "")
            tab[short][cname] = total / n
            tab[short]["dispatches"] = n
            tab[short]["dur_us"] = dur / n / 1e3
    print(f"{'kernel':52s} {'disp':>5s} {'dur_us':>8s} " + " ".join(f"{c:>18s}" for c in counters) + "   (per dispatch; dur under counter collection)")
    for k, v in sorted(tab.items(), key=lambda kv: -kv[1].get(counters[0], 0.0) * kv[1]["dispatches"]):
        print(f"{k:52s} {v['dispatches']:5d} {v['dur_us']:8.1f} " + " ".join(f"{v.get(c, float('nan')):18.1f}" for c in counters))


if __name__ == "__main__":
    main()
