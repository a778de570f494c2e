"""TEST INFRASTRUCTURE ONLY.  Builds tests/cpu_product/_build/libmifx_cpu.so: the product's host sources (diligentfx_amd/csrc/*.cpp, exactly as they ship) linked with
stub_device.cpp instead of the kernels and the HIP runtime.  See stub_device.cpp."""
import concurrent.futures
import glob
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libmifx_cpu.so")


def build():
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        return None
    csrc = os.path.join(ROOT, "diligentfx_amd", "csrc")
    srcs = sorted(glob.glob(os.path.join(csrc, "*.cpp"))) + [os.path.join(HERE, "stub_device.cpp")]
    deps = srcs + sorted(glob.glob(os.path.join(csrc, "*.h"))) + [os.path.join(ROOT, "include", "mifx.h")]
    h = hashlib.sha256()
    for d in deps:
        h.update(open(d, "rb").read())
    stamp = h.hexdigest()
    if os.path.exists(OUT) and os.path.exists(OUT + ".stamp") and open(OUT + ".stamp").read() == stamp:
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)

    def cc(src):
        obj = os.path.join(OUT_DIR, os.path.basename(src) + ".o")
        r = subprocess.run([hipcc, "-O1", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-I", os.path.join(ROOT, "include"), "-I", csrc, "-c", src, "-o", obj], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-3000:]
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        objs = list(ex.map(cc, srcs))
    r = subprocess.run(["g++", "-shared", "-o", OUT] + objs + ["-Wl,--no-undefined", "-ldl", "-lpthread"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    open(OUT + ".stamp", "w").write(stamp)
    return OUT


if __name__ == "__main__":
    print(build())
