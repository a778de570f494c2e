#!/usr/bin/env python3
"""libmifx_asan.so: the HOST objects of the library (csrc/*.cpp: the C ABI, the host sequencing of the effects, the communicator) compiled with
AddressSanitizer + UndefinedBehaviorSanitizer and linked with the regular device objects (diligentfx_amd/build/*.hip.o) -- the analogue of the
reference's sanitizer CI jobs (.github/workflows/build-linux.yml:57-69, ASAN / TSAN builds of the host code).

    python tools/build_sanitized.py            -> prints the library path and the sanitizer runtime to LD_PRELOAD
    LD_PRELOAD=<runtime> MIFX_LIB_PATH=<lib> ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_abi.py tests/test_comm.py -m "not gpu"

(tests/test_sanitizers.py does both.)  Device code is not instrumented: a sanitizer for gfx950 kernels is a different tool (compute-sanitizer has no ROCm
counterpart in this image)."""
import concurrent.futures
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import build as B  # noqa: E402

OUT = os.path.join(B.HERE, "libmifx_asan.so")
OBJDIR = os.path.join(B.OBJDIR, "asan")
SAN = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined", "-g", "-O1", "-shared-libsan"]


def runtime():
    return subprocess.run(["/opt/rocm/lib/llvm/bin/clang++", "-print-file-name=libclang_rt.asan-x86_64.so"], capture_output=True, text=True, check=True).stdout.strip()


def build():
    B.build()  # the device objects (and the regular library) must exist
    os.makedirs(OBJDIR, exist_ok=True)
    cc = B.hipcc()
    flags = [f for f in B.HIPCC_FLAGS if f != "-O3"] + SAN

    def one(src):
        obj = os.path.join(OBJDIR, os.path.basename(src) + ".o")
        if os.path.exists(obj) and os.path.getmtime(obj) > max(os.path.getmtime(p) for p in [src] + glob.glob(os.path.join(B.CSRC, "*.h")) + [os.path.join(ROOT, "include", "mifx.h")]):
            return obj
        r = subprocess.run([cc] + flags + ["-c", src, "-o", obj], capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError(f"sanitizer build failed on {os.path.basename(src)}")
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
        host = list(ex.map(one, sorted(glob.glob(os.path.join(B.CSRC, "*.cpp")))))
    dev = [os.path.join(B.OBJDIR, os.path.basename(s) + ".o") for s in sorted(glob.glob(os.path.join(B.CSRC, "*.hip")))]  # (not a glob: objects of removed sources may linger)
    r = subprocess.run([cc, f"--offload-arch={B.ARCH}", "-shared", "-fPIC"] + SAN + ["-o", OUT] + host + dev, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(r.stdout + r.stderr)
        raise RuntimeError("linking libmifx_asan.so failed")
    return OUT


if __name__ == "__main__":
    print(build())
    print(runtime())
