#!/usr/bin/env python3
"""How many kernels of the chain run at once?  (rocprofv3 --kernel-trace results database of a bench run with mifx_chain_set_overlap 1 / 2)

    python tools/overlap_stats.py <dir> [seconds-to-skip-at-both-ends]

Prints the share of the steady-state window with 0, 1, 2, 3+ kernels in flight, and per kernel: launches, average duration, and the share of its run time during which it
was the only kernel on the GPU."""
import collections
import glob
import os
import sqlite3
import sys

db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0])
rows = db.execute("select name, start, end from kernels where name like '%mifx::%' and name not like '%ibl_%' order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void mifx::", "").replace("mifx::", "")[:44]
t0, t1 = rows[0][1], max(e for _, _, e in rows)
lo, hi = t0 + (t1 - t0) * 0.35, t0 + (t1 - t0) * 0.95  # the timed region of the bench run (the warm-up and the sweep come first)
ev = []
for i, (n, s, e) in enumerate(rows):
    if e <= lo or s >= hi:
        continue
    ev.append((max(s, lo), 1, i))
    ev.append((min(e, hi), -1, i))
ev.sort()
active, last = set(), lo
depth_time = collections.Counter()
alone = collections.Counter()
for t, d, i in ev:
    dt = t - last
    if dt > 0:
        depth_time[min(len(active), 3)] += dt
        if len(active) == 1:
            alone[short(rows[next(iter(active))][0])] += dt
    last = t
    if d > 0:
        active.add(i)
    else:
        active.discard(i)
span = hi - lo
print(f"window {span / 1e6:.2f} ms; kernels in flight:  " + "  ".join(f"{k if k < 3 else '3+'}: {100.0 * v / span:.1f} %" for k, v in sorted(depth_time.items())))
tot = collections.defaultdict(lambda: [0, 0.0])
for n, s, e in rows:
    if s >= lo and e <= hi:
        tot[short(n)][0] += 1
        tot[short(n)][1] += e - s
print(f"{'kernel':46s} {'launches':>8s} {'avg_us':>8s} {'alone %':>8s}")
for k, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:46s} {c:8d} {d / c / 1e3:8.1f} {100.0 * alone[k] / d:8.1f}")
