#!/usr/bin/env python3
"""Dependent memory round trips of a kernel, read off its gfx950 assembly (round 5): the kernels of the chain are latency-bound at their occupancy -- time = waves x
(dependent round trips x loaded latency) / resident waves -- so a load whose first use (s_waitcnt vmcnt) follows within a few instructions, with nothing else in flight,
is a round trip nothing hides.  Prints, per kernel, the sequence  L<n> = n vector-memory loads issued, W<k>(+d) = s_waitcnt vmcnt(k) d instructions after the last load,
B = barrier, loop / branch markers -- and flags the exposed ones (d <= 12 and k == 0).
    ISA_KEEP=/tmp/isa python tools/isa_stats.py <name>   (writes the .s files)      python tools/isa_roundtrips.py /tmp/isa/<file>.s <mangled-name substring>"""
import re
import sys

path, want = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
out, inside, since, nload = [], False, 0, 0
for ln in lines:
    if not inside:
        if re.match(r"^_ZN\S*" + re.escape(want) + r"\S*:", ln):
            inside, name = True, ln.split(":")[0]
            seq, since, nload, exposed = [], 0, 0, 0
        continue
    t = ln.strip()
    if t.startswith("s_endpgm"):
        if nload:
            seq.append(f"L{nload}")
        print(name[:110])
        print("   " + " ".join(seq))
        print(f"   exposed round trips (a full wait within 12 instructions of the last load): {exposed}")
        inside = False
        continue
    if not t or t.startswith(";") or t.startswith("."):
        if t.startswith(".LBB"):
            if nload:
                seq.append(f"L{nload}"); nload = 0
            seq.append("|")
        continue
    op = t.split()[0]
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        nload += 1
        since = 0
        continue
    since += 1
    m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", t)
    if m:
        if nload:
            seq.append(f"L{nload}"); nload = 0
        k = int(m.group(1))
        flag = "!" if (k == 0 and since <= 12) else ""
        exposed += 1 if flag else 0
        seq.append(f"W{k}(+{since}){flag}")
    elif op == "s_barrier":
        if nload:
            seq.append(f"L{nload}"); nload = 0
        seq.append("B")
    elif op.startswith("s_cbranch"):
        if nload:
            seq.append(f"L{nload}"); nload = 0
        seq.append("br")
