#!/usr/bin/env python3
"""oracle/build.py -- TEST INFRASTRUCTURE build recipe (the checker, not the product).

  build_oracle(): g++ oracle/mifx_oracle.cpp  -> oracle/libmifx_oracle.so   (hand-written CPU restatement, "port")
  build_ref():    /root/reference shader source, compiled where it lies through oracle/ref/hlsl_shim.h
                  -> oracle/_ref/libmifx_ref.so   (the reference itself on the CPU, "reference").
                  Only possible where /root/reference exists (this container); the .so travels to the GPU box.
"""
import concurrent.futures
import glob
import hashlib
import os
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE_ROOT = os.environ.get("MIFX_REFERENCE_ROOT", "/root/reference")
CXXFLAGS = ["-O2", "-fPIC", "-fopenmp", "-fsingle-precision-constant", "-ffp-contract=off", "-fno-fast-math", "-w"]


def _run(cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        sys.stderr.write(" ".join(cmd) + "\n" + r.stdout + r.stderr + "\n")
        raise RuntimeError("command failed: " + " ".join(cmd[:3]) + " ...")
    return r


def _stamp(paths, extra=""):
    h = hashlib.sha1(extra.encode())
    for p in sorted(paths):
        h.update(p.encode())
        with open(p, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def _up_to_date(out, stamp):
    sp = out + ".stamp"
    return os.path.exists(out) and os.path.exists(sp) and open(sp).read() == stamp


def build_oracle(force=False):
    units = [os.path.join(HERE, "mifx_oracle.cpp")] + sorted(glob.glob(os.path.join(HERE, "oracle_*.cpp")))
    src = units + glob.glob(os.path.join(HERE, "*.h"))
    out = os.path.join(HERE, "libmifx_oracle.so")
    stamp = _stamp(src, " ".join(CXXFLAGS))
    if not force and _up_to_date(out, stamp):
        return out
    _run(["g++", "-std=c++17", "-shared", "-march=x86-64-v2"] + CXXFLAGS + ["-o", out] + units)
    open(out + ".stamp", "w").write(stamp)
    return out


def build_ref(force=False):
    """Returns the path of libmifx_ref.so, or None when the reference tree is not available."""
    outdir = os.path.join(HERE, "_ref")
    out = os.path.join(outdir, "libmifx_ref.so")
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "Shaders")):
        return out if os.path.exists(out) else None
    os.makedirs(outdir, exist_ok=True)
    wrappers = sorted(glob.glob(os.path.join(HERE, "ref", "ref_*.cpp")))
    deps = wrappers + glob.glob(os.path.join(HERE, "ref", "*.h")) + glob.glob(os.path.join(HERE, "ref", "*.inc")) + glob.glob(os.path.join(HERE, "*.h")) + [os.path.join(HERE, "ref_prep.py")]
    stamp = _stamp(deps, " ".join(CXXFLAGS))
    if not force and _up_to_date(out, stamp):
        return out
    sys.path.insert(0, HERE)
    import ref_prep

    with tempfile.TemporaryDirectory(prefix="mifx_ref_") as tmp:
        if ref_prep.main(REFERENCE_ROOT, tmp) != 0:
            raise RuntimeError("ref_prep failed")

        def cc(src):
            obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
            _run(["g++", "-std=c++20", "-c"] + CXXFLAGS + ["-I", os.path.join(HERE, "ref"), "-I", tmp, "-o", obj, src])
            return obj

        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            objs = list(ex.map(cc, wrappers))
        _run(["g++", "-shared", "-fopenmp", "-o", out] + objs)
    open(out + ".stamp", "w").write(stamp)
    return out


REFHOST_SOURCES = [  # the reference's host classes of the hot path, compiled where they lie
    "PostProcess/Common/src/PostFXContext.cpp",
    "PostProcess/Common/src/PostFXRenderTechnique.cpp",
    "PostProcess/ScreenSpaceAmbientOcclusion/src/ScreenSpaceAmbientOcclusion.cpp",
    "PostProcess/ScreenSpaceReflection/src/ScreenSpaceReflection.cpp",
    "PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp",
    "PostProcess/Bloom/src/Bloom.cpp",
    "PostProcess/DepthOfField/src/DepthOfField.cpp",
    "Components/src/EnvMapRenderer.cpp",
]


def build_refhost(force=False):
    """oracle/_ref/libmifx_refhost.so: the reference's HOST code (PostFXContext, SSAO, SSR, TAA, Bloom, DepthOfField) compiled from /root/reference against the recording DiligentCore
    stand-in oracle/refhost/dg (DiligentCore itself is not part of the reference tree).  Returns its path, or None when the reference tree is not available and no
    prebuilt library travelled."""
    outdir = os.path.join(HERE, "_ref")
    out = os.path.join(outdir, "libmifx_refhost.so")
    if not os.path.isdir(os.path.join(REFERENCE_ROOT, "PostProcess")):
        return out if os.path.exists(out) else None
    os.makedirs(outdir, exist_ok=True)
    rh = os.path.join(HERE, "refhost")
    dg = os.path.join(rh, "dg")
    own = [os.path.join(rh, "refhost.cpp")]
    ref = [os.path.join(REFERENCE_ROOT, s) for s in REFHOST_SOURCES]
    deps = own + ref + glob.glob(os.path.join(dg, "flat", "*")) + glob.glob(os.path.join(REFERENCE_ROOT, "PostProcess", "*", "interface", "*.hpp")) + \
        glob.glob(os.path.join(REFERENCE_ROOT, "Components", "interface", "EnvMapRenderer.hpp"))
    flags = ["-std=c++17", "-O1", "-fPIC", "-w"]
    stamp = _stamp(deps, " ".join(flags))
    if not force and _up_to_date(out, stamp):
        return out
    # "../../../../DiligentCore/..." (how the reference's headers reach DiligentCore) resolves against an include directory four levels below dg/
    inc = ["-I", os.path.join(dg, "anchor", "a", "b", "c"), "-I", os.path.join(dg, "anchor", "a", "b"), "-I", os.path.join(dg, "flat"), "-I", REFERENCE_ROOT, "-I", os.path.join(REFERENCE_ROOT, "PostProcess", "Common", "interface"),
           "-I", os.path.join(REFERENCE_ROOT, "PostProcess", "Common", "src")]
    for e in ("ScreenSpaceAmbientOcclusion", "ScreenSpaceReflection", "TemporalAntiAliasing", "Bloom", "DepthOfField"):
        inc += ["-I", os.path.join(REFERENCE_ROOT, "PostProcess", e, "interface")]
    inc += ["-I", os.path.join(REFERENCE_ROOT, "Components", "interface")]
    with tempfile.TemporaryDirectory(prefix="mifx_refhost_") as tmp:

        def cc(src):
            obj = os.path.join(tmp, os.path.basename(src)[:-4] + ".o")
            _run(["g++", "-c"] + flags + inc + ["-o", obj, src])
            return obj

        with concurrent.futures.ThreadPoolExecutor(max_workers=8) as ex:
            objs = list(ex.map(cc, own + ref))
        _run(["g++", "-shared", "-o", out] + objs)
    open(out + ".stamp", "w").write(stamp)
    return out


if __name__ == "__main__":
    force = "--force" in sys.argv
    if os.path.exists(os.path.join(HERE, "mifx_oracle.cpp")):
        print(build_oracle(force))
    print(build_ref(force))
    print(build_refhost(force))
