#!/usr/bin/env python3
"""Which kernels changed since a revision?  Compiles every diligentfx_amd/csrc/*.hip of <rev> (a temporary git worktree) and of the working tree to gfx950 assembly with
the build's flags and compares the kernels instruction for instruction (comments and block numbers stripped).  What a GPU test run of <rev> still says about the working
tree: everything whose kernels are identical.  No GPU needed.

    python tools/isa_identity.py 178790a > profiles/r04_isa_identity_v19_vs_final.txt"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "-fvisibility=hidden", "-fno-slp-vectorize", "--cuda-device-only", "-S", "-x", "hip"]


def kernels(asm_path):
    out, cur = {}, None
    for line in open(asm_path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            cur = out.setdefault(m.group(1), [])
            continue
        if cur is not None:
            if line.startswith(".Lfunc_end"):
                cur = None
                continue
            text = re.sub(r"\.LBB\d+_", ".LBB_", re.sub(r";.*", "", line).strip())
            if text:
                cur.append(text)
    return out


def assemble(tree, src, out):
    first = open(src).readline()
    extra = first.split("MIFX_BUILD_FLAGS:")[1].split() if "MIFX_BUILD_FLAGS:" in first else []
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + extra + ["-I", os.path.join(tree, "include"), "-I", os.path.join(tree, "diligentfx_amd", "csrc"), src, "-o", out], capture_output=True, text=True)
    return r.returncode == 0


def main():
    rev = sys.argv[1]
    with tempfile.TemporaryDirectory(prefix="mifx_isa_") as tmp:
        old = os.path.join(tmp, "old")
        subprocess.run(["git", "-C", ROOT, "worktree", "add", "-f", "-q", old, rev], check=True)
        try:
            print(f"kernels of the working tree ({subprocess.run(['git', '-C', ROOT, 'rev-parse', '--short', 'HEAD'], capture_output=True, text=True).stdout.strip()} + changes) against {rev}, "
                  "gfx950 assembly, the build's flags")
            for src in sorted(glob.glob(os.path.join(ROOT, "diligentfx_amd", "csrc", "*.hip"))):
                name = os.path.basename(src)
                prev = os.path.join(old, "diligentfx_amd", "csrc", name)
                if not os.path.exists(prev):
                    print(f"{name}: new file")
                    continue
                a, b = os.path.join(tmp, name + ".old.s"), os.path.join(tmp, name + ".new.s")
                if not (assemble(old, prev, a) and assemble(ROOT, src, b)):
                    print(f"{name}: did not compile")
                    continue
                ka, kb = kernels(a), kernels(b)
                diff = [k for k in ka if k in kb and ka[k] != kb[k]]
                print(f"{name}: {sum(1 for k in ka if k in kb and ka[k] == kb[k])} kernels identical, {len(diff)} different, {len(set(kb) - set(ka))} new, {len(set(ka) - set(kb))} gone")
                for k in diff:
                    print("    different:", subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()[:160] or k)
        finally:
            subprocess.run(["git", "-C", ROOT, "worktree", "remove", "--force", old])


if __name__ == "__main__":
    main()
