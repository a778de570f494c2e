"""The host side of the library under AddressSanitizer + UndefinedBehaviorSanitizer (the reference's CI builds its host code with ASAN / TSAN,
.github/workflows/build-linux.yml:57-69): tools/build_sanitized.py links sanitized host objects with the regular device objects, and the CPU-side
ABI / communicator tests run against that library in a child process with the sanitizer runtime preloaded.  Any report fails the child."""
import os
import subprocess
import sys

from util import ROOT


def test_host_objects_under_asan_ubsan():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import build_sanitized

    lib = build_sanitized.build()
    env = dict(os.environ, MIFX_LIB_PATH=lib, LD_PRELOAD=build_sanitized.runtime(), ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_abi.py", "tests/test_comm.py", "-m", "not gpu", "-q", "-x", "-p", "no:cacheprovider"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "passed" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
    assert "ERROR: AddressSanitizer" not in r.stderr and "runtime error:" not in r.stderr, r.stderr[-3000:]
