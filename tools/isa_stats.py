#!/usr/bin/env python3
"""Static gfx950 instruction mix per kernel (hipcc -S on each csrc/*.hip): total / VALU / packed / division expansion / transcendental / memory.

    python tools/isa_stats.py [substring-of-kernel-name ...]"""
import collections
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import build as B  # noqa: E402


def classify(op, c):
    c["total"] += 1
    if op.startswith("v_"):
        c["valu"] += 1
    if op.startswith("s_"):
        c["salu"] += 1
    if op.startswith("v_pk_"):
        c["pk"] += 1
    if op.startswith("v_div_"):
        c["div*"] += 1
    if op.startswith(("v_fma", "v_fmac")):
        c["fma"] += 1
    if op.startswith(("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")):
        c["trans"] += 1
    if op.startswith(("global_load", "buffer_load")):
        c["vmem_ld"] += 1
    if op.startswith(("global_store", "buffer_store")):
        c["vmem_st"] += 1
    if op.startswith("ds_"):
        c["lds"] += 1
    if op.startswith(("s_cbranch", "s_branch")):
        c["branch"] += 1


def main():
    want = sys.argv[1:]
    with tempfile.TemporaryDirectory() as tmp:
        for src in sorted(glob.glob(os.path.join(B.CSRC, "*.hip"))):
            extra = []
            first = open(src).readline()
            if first.startswith("// MIFX_BUILD_FLAGS:"):
                extra = first.split(":", 1)[1].split()
            asm = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.run([B.hipcc()] + B.HIPCC_FLAGS + extra + ["-x", "hip", "--cuda-device-only", "-S", src, "-o", asm], check=True, capture_output=True)
            cur, stats = None, {}
            for line in open(asm):
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    cur = m.group(1)
                    stats[cur] = collections.Counter()
                elif cur and line.startswith("\t") and not line.startswith(("\t.", "\t;")):
                    classify(line.split()[0], stats[cur])
            for k, c in stats.items():
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void mifx::", "")
                if c["total"] < 20 or (want and not any(w in name for w in want)):
                    continue
                print(f"{name:48s} " + " ".join(f"{f}={c[f]}" for f in ("total", "valu", "salu", "pk", "div*", "fma", "trans", "vmem_ld", "vmem_st", "lds", "branch")))


if __name__ == "__main__":
    main()
