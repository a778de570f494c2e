#!/usr/bin/env python3
"""Per-kernel average duration grouped by launch grid (rocprofv3 --kernel-trace results database): separates whole-frame launches from
row-band launches of the same kernel.   python tools/grid_stats.py <dir> [min_calls]"""
import glob
import os
import sqlite3
import sys

db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0])
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rows = db.execute("select name, grid_x, grid_y, workgroup_x, workgroup_y, count(*), avg(duration) from kernels where name like '%mifx::%' "
                  "group by name, grid_x, grid_y order by name, grid_y desc").fetchall()
print(f"{'kernel':44s} {'blocks x':>9s} {'blocks y':>9s} {'calls':>6s} {'avg_us':>9s}")
for name, gx, gy, wx, wy, n, dur in rows:
    if n >= min_calls and "ibl_" not in name:
        short = name.split("(")[0].replace("void mifx::", "").replace("mifx::", "")[:44]
        print(f"{short:44s} {gx // max(wx, 1):9d} {gy // max(wy, 1):9d} {n:6d} {dur / 1e3:9.1f}")
tot = db.execute("select count(*), sum(duration), min(start), max(end) from kernels where name like '%mifx::%' and name not like '%ibl_%'").fetchone()
print(f"all mifx kernels: {tot[0]} launches, busy {tot[1] / 1e6:.3f} ms, span {(tot[3] - tot[2]) / 1e6:.3f} ms")
