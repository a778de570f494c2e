"""TEST INFRASTRUCTURE ONLY.  Prints the launcher stubs of stub_device.cpp (the block below its "the launchers" marker) from the declarations of
diligentfx_amd/csrc/mifx_host.h: one definition per `mifx_status launch_*(...)`, which hands its arguments to record().  Re-run and paste when a launcher's signature changes
(the link of tests/cpu_product/build.py fails with the launcher's name until then: -Wl,--no-undefined).

    python tests/cpu_product/gen_stubs.py > /tmp/stubs.inc"""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    src = open(os.path.join(ROOT, "diligentfx_amd", "csrc", "mifx_host.h")).read()
    for d in re.findall(r"\n(mifx_status\s+launch_\w+\s*\((?:[^;]|\n)*?\))\s*;", src):
        d = " ".join(re.sub(r"//[^\n]*", "", d).split())
        m = re.match(r"mifx_status (launch_\w+)\s*\((.*)\)$", d)
        if not m:
            continue
        name, params = m.group(1), m.group(2)
        parts, depth, cur = [], 0, ""
        for ch in params:
            depth += ch in "(<[{"
            depth -= ch in ")>]}"
            if ch == "," and depth == 0:
                parts.append(cur.strip())
                cur = ""
            else:
                cur += ch
        if cur.strip():
            parts.append(cur.strip())
        decl, names = [], []
        for p in parts:
            p = re.sub(r"\s*=\s*.+$", "", p)  # default arguments belong to the declaration
            decl.append(p)
            names.append(re.match(r"(.*?)(\w+)(\[\d*\])?$", p).group(2))
        args = ", ".join(n for n in names if n != "s")
        print(f"mifx_status {name}({', '.join(decl)})\n{{\n    t_stream = s;\n    return record(\"{name[7:]}\", {args});\n}}")


if __name__ == "__main__":
    main()
