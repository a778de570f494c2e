#!/usr/bin/env python3
"""Prints VGPR / SGPR / scratch / LDS / occupancy of every gfx950 kernel in diligentfx_amd/csrc (hipcc -Rpass-analysis=kernel-resource-usage).
`collect()` returns the same as a list of dicts (tests/test_kernel_resources.py: no kernel of the shipped build may spill)."""
import concurrent.futures
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import build as B  # noqa: E402


def _one(src):
    extra = []
    first = open(src).readline()
    if first.startswith("// MIFX_BUILD_FLAGS:"):
        extra = first.split(":", 1)[1].split()
    if os.path.basename(src) in B.fma_sources():
        extra += ["-ffp-contract=" + B.FMA_MODE.get(os.path.basename(src), "fast")]
    r = subprocess.run([B.hipcc()] + B.HIPCC_FLAGS + extra + ["-x", "hip", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True)
    out, cur = [], {}
    for line in r.stderr.splitlines():
        m = re.search(r"remark: .*?(Function Name|VGPRs|AGPRs|SGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        k, v = m.group(1).split(" [")[0], m.group(2).strip()
        if k == "Function Name":
            cur = {"name": v, "file": os.path.basename(src)}
        else:
            cur[k] = v
        if k == "LDS Size":
            cur["demangled"] = subprocess.run(["c++filt", cur["name"]], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void mifx::", "")
            out.append(cur)
    return out


def collect():
    srcs = sorted(glob.glob(os.path.join(B.CSRC, "*.hip")))
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        return [k for ks in ex.map(_one, srcs) for k in ks]


def workgroups_per_cu(k, threads=256):
    """Workgroups of `threads` threads a CU admits: what the register allocation allows (Occupancy = waves per SIMD, four SIMDs), the LDS (160 KiB per CU), the hardware's
    eight -- and the scalar registers: min(8, 800 / (16 ceil(sgprs / 16) + 16)) for 256-thread workgroups (MI355X_MICROARCH.md "Residency": <= 80 SGPRs -> 8, 82 - 96 -> 7, 98 -> 6),
    which the `Occupancy` remark does not know.  Round 5: SSR's ray march had 84 - 88 SGPRs and so ran at seven workgroups per CU with "occ 8"."""
    waves = threads // 64
    by_regs = int(k.get("Occupancy", "8")) * 4 // waves
    lds = int(k.get("LDS Size", "0"))
    by_lds = (160 * 1024) // lds if lds > 0 else 99
    sgprs = int(k.get("SGPRs", "0"))
    by_sgprs = 800 // (((sgprs + 15) // 16) * 16 + 16) if threads == 256 else 99
    return min(8 * 4 // waves, by_regs, by_lds, by_sgprs)


if __name__ == "__main__":
    for k in collect():
        print(f"{k['file']:16s} {k['demangled']:52s} vgpr {k.get('VGPRs'):>4s} sgpr {k.get('SGPRs'):>4s} scratch {k.get('ScratchSize'):>5s} occ {k.get('Occupancy'):>2s} lds {k.get('LDS Size'):>6s}"
              f"  wg/CU(256 thr) {workgroups_per_cu(k)}")
