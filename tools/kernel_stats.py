#!/usr/bin/env python3
"""Prints the mifx kernels of a rocprofv3 --kernel-trace --stats results database (sqlite) as a table: python tools/kernel_stats.py <dir-or-db> [title]"""
import glob
import os
import sqlite3
import sys


def main():
    path = sys.argv[1]
    title = sys.argv[2] if len(sys.argv) > 2 else ""
    dbs = [path] if path.endswith(".db") else glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True)
    db = sqlite3.connect(dbs[0])
    print(title)
    print(f"{'kernel':64s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s}")
    tot = 0.0
    for name, calls, total, avg, pct in db.execute("select * from top_kernels"):
        if "mifx::" in name and "ibl_" not in name:
            short = name.split("(")[0].replace("void ", "")
            print(f"{short:64s} {calls:6d} {total:12.1f} {avg:10.2f}")
            tot += total
    print(f"{'sum of mifx chain kernels':64s} {'':6s} {tot:12.1f}")


if __name__ == "__main__":
    main()
