#!/usr/bin/env python3
"""Idle time between kernels of the chain (rocprofv3 --kernel-trace results database): the device is idle whenever no mifx kernel is running between the first
and the last launch of a frame.   python tools/gap_stats.py <dir> [frames-to-skip]

Prints, per steady-state frame: span, busy (union of the kernel intervals), idle = span - busy, and the number of launches; then the largest idle gaps with the
kernels on either side."""
import glob
import os
import sqlite3
import sys

db = sqlite3.connect(glob.glob(os.path.join(sys.argv[1], "**", "*_results.db"), recursive=True)[0])
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rows = db.execute("select name, start, end from kernels where name like '%mifx::%' and name not like '%ibl_%' order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void mifx::", "").replace("mifx::", "")[:40]
# a frame starts at its cube_apron / blue_noise launch burst: split at the tone-map / final Bloom kernel, the last launch of a frame
frames, cur = [], []
for name, s, e in rows:
    cur.append((short(name), s, e))
    if "bloom_final_tonemap" in name or ("tonemap_kernel" in name and "bloom" not in name):
        frames.append(cur)
        cur = []
frames = frames[skip:]
if not frames:
    sys.exit("no complete frames in the trace")
tot_span = tot_busy = 0.0
gaps = []
for f in frames:
    span = (max(e for _, _, e in f) - f[0][1]) / 1e3
    busy, hi, last = 0.0, f[0][1], f[0][0]
    for n, s, e in f:
        if s > hi:
            gaps.append(((s - hi) / 1e3, last, n))
            busy += (e - s) / 1e3
        else:
            busy += max(0, e - hi) / 1e3
        if e > hi:
            hi, last = e, n
    tot_span += span
    tot_busy += busy
n = len(frames)
print(f"{n} frames: span {tot_span / n:.1f} us, busy {tot_busy / n:.1f} us, idle {(tot_span - tot_busy) / n:.1f} us per frame, {sum(len(f) for f in frames) / n:.0f} launches per frame")
agg = {}
for g, a, b in gaps:
    k = (a, b)
    agg.setdefault(k, []).append(g)
print(f"{'after':40s} {'before':40s} {'count':>6s} {'avg_us':>8s} {'us/frame':>9s}")
for (a, b), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:40]:
    print(f"{a:40s} {b:40s} {len(v):6d} {sum(v) / len(v):8.2f} {sum(v) / n:9.2f}")
