#!/usr/bin/env python3
"""Static gfx950 instruction mix per kernel (hipcc -S on each csrc/*.hip): total / VALU / packed / division expansion / transcendental / memory.

    python tools/isa_stats.py [substring-of-kernel-name ...]        (ISA_SRC=<file substring> restricts the sources, ISA_EXTRA adds hipcc flags)

`cost` = the VALU instructions weighted by their measured issue cost (in units of one v_fma_f32): the chain's kernels are VALU-issue bound, so for
straight-line kernels it tracks the kernel time; ISA_KEEP=<dir> keeps the .s files."""
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


# Issue cost of a wave64 VALU instruction in units of one full-rate instruction (2 shader cycles on the SIMD-32 of CDNA4), measured on an MI355X over >= 50 ms per opcode
# with the shader clock recorded (tools/microbench/valu_rate3.hip, profiles/r03_valu_issue_rate.txt, 8 waves per SIMD):
#   1.0  (0.89 - 0.97 ns, 2.1 - 2.2 cycles)  v_fma / mul / add / mov / and / add_u32, also with a 32-bit literal
#   2.0  (1.69 ns, 4.05 cycles)              min / max / med3 / min3, conversions, floor / fract, shifts, v_lshl_add, integer multiplies, v_div_fixup, packed fp32
#                                            (two lanes of work), and ANY of the full-rate ones when a source is an SGPR
#   4.0  (3.39 ns, 8.1 cycles)               the transcendental unit (rcp / sqrt / rsq / exp / log / sin / cos)
#   a v_cmp + v_cndmask pair costs 3 (2.56 ns): priced 2 + 1
# (round 2's table had 1.4 and 2.5 for the last two classes, from 0.1 ms launches converted at the nominal clock.)
FULL_RATE = ("v_fma_f32", "v_fmac_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_cndmask_b32", "v_add_u32",
             "v_sub_u32", "v_subrev_u32")
TRANS = ("v_rcp", "v_sqrt", "v_rsq", "v_exp", "v_log", "v_sin", "v_cos")
SGPR_OPERAND = re.compile(r"(?<![\w])(s\d+|s\[\d+:\d+\]|ttmp\d+)(?![\w])")


def issue_cost(op, operands):
    if not op.startswith("v_"):
        return 0.0
    if op.startswith(TRANS):
        return 4.0
    base = op[:-4] if op.endswith(("_e32", "_e64")) else op
    if base in FULL_RATE:
        srcs = operands.split(",", 1)[1] if "," in operands else ""
        return 2.0 if SGPR_OPERAND.search(srcs) and base != "v_cndmask_b32" else 1.0
    return 2.0


def classify(op, c, operands=""):
    c["total"] += 1
    cost = issue_cost(op, operands)
    c["cost"] += cost
    if op.startswith("v_"):
        c["full" if cost == 1.0 else "trans4" if cost == 4.0 else "half"] += 1
        base = op[:-4] if op.endswith(("_e32", "_e64")) else op
        srcs = operands.split(",", 1)[1] if "," in operands else ""
        if base in FULL_RATE and base != "v_cndmask_b32" and SGPR_OPERAND.search(srcs):
            c["sgpr_src"] += 1
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
        if os.environ.get("ISA_KEEP"):
            tmp = os.environ["ISA_KEEP"]
            os.makedirs(tmp, exist_ok=True)
        only = os.environ.get("ISA_SRC")  # e.g. ISA_SRC=ssao_ao: compile that source only
        for src in sorted(glob.glob(os.path.join(B.CSRC, "*.hip"))):
            if only and only not in os.path.basename(src):
                continue
            extra = os.environ.get("ISA_EXTRA", "").split()
            first = open(src).readline()
            if first.startswith("// MIFX_BUILD_FLAGS:"):
                extra += first.split(":", 1)[1].split()
            if os.path.basename(src) in B.fma_sources():  # as the build compiles it
                extra += ["-ffp-contract=" + B.FMA_MODE.get(os.path.basename(src), "fast")]
            asm = os.path.join(tmp, os.path.basename(src) + ".s")
            subprocess.run([B.hipcc()] + B.HIPCC_FLAGS + extra + ["-x", "hip", "--cuda-device-only", "-S", src, "-o", asm], check=True, capture_output=True)
            cur, stats = None, {}
            for line in open(asm):
                m = re.match(r"^(_Z\w+):", line)
                if m:
                    cur = m.group(1)
                    stats[cur] = collections.Counter(); stats[cur]['cost'] = 0.0
                elif cur and line.startswith("\t") and not line.startswith(("\t.", "\t;")):
                    f = line.split(None, 1)
                    classify(f[0], stats[cur], f[1] if len(f) > 1 else "")
            for k, c in stats.items():
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().split("(")[0].replace("void mifx::", "")
                if c["total"] < 20 or (want and not any(w in name for w in want)):
                    continue
                print(f"{name:48s} " + " ".join(f"{f}={c[f]}" for f in ("total", "valu", "full", "half", "trans4", "sgpr_src", "salu", "vmem_ld", "vmem_st", "lds", "branch")) + f" cost={c['cost']:.0f}")


if __name__ == "__main__":
    main()
