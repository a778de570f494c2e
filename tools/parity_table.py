#!/usr/bin/env python3
"""Outlier fractions of the GPU parity suite, one row per comparison: the shipped build beside the strict build (exact division and square roots, no contraction,
MIFX_R4_STRICT) -- what every deliberate deviation from the reference's arithmetic costs, in the unit the tests decide in.

    MIFX_PARITY_LOG=/tmp/shipped.jsonl python -m pytest tests -m gpu -q
    MIFX_FMA_SOURCES=none: build_variant(..., ['-DMIFX_PRECISE_MATH=1', '-DMIFX_R4_STRICT=1']) -> MIFX_LIB_PATH=.../strict.so MIFX_PARITY_LOG=/tmp/strict.jsonl MIFX_PARITY_MEASURE=1 python -m pytest ...
    python tools/parity_table.py /tmp/shipped.jsonl /tmp/strict.jsonl

Comparisons that differ only in a frame number are folded into one row (the worst frame)."""
import collections
import json
import re
import sys


def load(path):
    tab = collections.OrderedDict()
    for line in open(path):
        r = json.loads(line)
        key = re.sub(r"\s*(frame|level|mip|case|rank)\s*\d+", "", r["what"]).strip()
        t = tab.setdefault(key, {"frac": 0.0, "allowed": r["allowed"], "n": 0, "count": 0})
        t["frac"] = max(t["frac"], r["frac"])
        t["allowed"] = max(t["allowed"], r["allowed"])
        t["n"] = max(t["n"], r["n"])
        t["count"] += 1
    return tab


def main():
    shipped, strict = load(sys.argv[1]), load(sys.argv[2])
    print(f"{'comparison (worst frame)':70s} {'values':>9s} {'budget':>9s} {'shipped':>10s} {'strict':>10s}")
    n_zero = 0
    for k, v in shipped.items():
        s = strict.get(k)
        if v["frac"] == 0.0 and (s is None or s["frac"] == 0.0):
            n_zero += 1
            continue
        print(f"{k[:70]:70s} {v['n']:9d} {v['allowed']:9.1e} {v['frac']:10.2e} " + (f"{s['frac']:10.2e}" if s else f"{'-':>10s}"))
    print(f"({n_zero} comparisons without a single value beyond rtol in either build are not listed; {len(shipped)} comparisons in all)")


if __name__ == "__main__":
    main()
