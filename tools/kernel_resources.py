#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / LDS / occupancy of every gfx950 kernel in diligentfx_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage)."""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import build as B  # noqa: E402

for src in sorted(glob.glob(os.path.join(B.CSRC, "*.hip"))):
    r = subprocess.run([B.hipcc()] + B.HIPCC_FLAGS + ["-x", "hip", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    cur = {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (.*)", line)
        if not m:
            continue
        k, v = m.group(1).split(" [")[0], m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v}
        else:
            cur[k] = v
        if k == "LDS Size":
            name = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void mifx::", "")
            print(f"{os.path.basename(src):16s} {name:52s} vgpr {cur.get('VGPRs'):>4s} sgpr {cur.get('SGPRs'):>4s} scratch {cur.get('ScratchSize'):>5s} occ {cur.get('Occupancy'):>2s} lds {cur.get('LDS Size'):>6s}")
