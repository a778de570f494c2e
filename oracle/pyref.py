"""oracle/pyref.py -- TEST INFRASTRUCTURE: ctypes binding of oracle/_ref/libmifx_ref.so (the reference's own
shader source compiled for the CPU) and of oracle/libmifx_oracle.so (the hand-written restatement).
Both libraries share one calling convention (`ref_args`, oracle/ref/ref_common.h).  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF_MAX_IN = 14
REF_MAX_MIPS = 12


class RefImg(ctypes.Structure):
    _fields_ = [("data", ctypes.c_void_p), ("w", ctypes.c_int), ("h", ctypes.c_int), ("c", ctypes.c_int)]


class RefArgs(ctypes.Structure):
    _fields_ = [
        ("inp", (RefImg * REF_MAX_MIPS) * REF_MAX_IN),
        ("in_mips", ctypes.c_int * REF_MAX_IN),
        ("out", RefImg * 4),
        ("cam0", ctypes.c_void_p),
        ("cam1", ctypes.c_void_p),
        ("attribs", ctypes.c_void_p),
        ("ival", ctypes.c_int * 8),
        ("fval", ctypes.c_float * 8),
    ]


def _img(a: np.ndarray) -> RefImg:
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    if a.ndim == 2:
        h, w = a.shape
        c = 1
    else:
        h, w, c = a.shape
    return RefImg(a.ctypes.data, w, h, c)


class _Lib:
    def __init__(self, path):
        self.path = path
        self.lib = ctypes.CDLL(path)

    def has(self, name):
        return hasattr(self.lib, name)

    def call(self, name, ins=(), outs=(), cam0=None, cam1=None, attribs=None, ival=(), fval=()):
        """ins: list of np.float32 arrays (H,W[,C]) or lists of arrays (mip chains); outs: list of arrays, written in place."""
        args = RefArgs()
        keep = []
        for i, t in enumerate(ins):
            if t is None:
                args.in_mips[i] = 0
                continue
            mips = t if isinstance(t, (list, tuple)) else [t]
            args.in_mips[i] = len(mips)
            for m, a in enumerate(mips):
                args.inp[i][m] = _img(a)
                keep.append(a)
        for i, a in enumerate(outs):
            args.out[i] = _img(a)
        for nm, v in (("cam0", cam0), ("cam1", cam1), ("attribs", attribs)):
            if v is not None:
                b = v if isinstance(v, (bytes, bytearray)) else bytes(v)
                # zero padding behind the block: attribute structs that grew a trailing field (PBRShadeAttribs::Workflow) are read whole by the checkers,
                # and fixtures recorded before the field existed then read 0 = the previous behaviour
                buf = ctypes.create_string_buffer(b, max(len(b), 2048))
                keep.append(buf)
                setattr(args, nm, ctypes.cast(buf, ctypes.c_void_p).value)
        for i, v in enumerate(ival):
            args.ival[i] = int(v)
        for i, v in enumerate(fval):
            args.fval[i] = float(v)
        fn = getattr(self.lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [ctypes.POINTER(RefArgs)]
        rc = fn(ctypes.byref(args))
        if rc != 0:
            raise RuntimeError(f"{os.path.basename(self.path)}:{name} returned {rc}")
        return outs


class QuantizingLib:
    """TEST INFRASTRUCTURE: a checker library whose passes store their 4-channel images into RGBA16_FLOAT targets -- every (H, W, 4) output is rounded to
    nearest-even binary16 after the call (numpy's IEEE float16), the format-emulation mode of SURVEY.md section 0.2 for the targets the reference keeps as
    RGBA16_FLOAT.  Used against the RGBA16_FLOAT storage build of the product library (libmifx_h4.so); cube maps are produced with the plain library."""

    def __init__(self, lib):
        self.lib = lib
        self.path = lib.path

    def has(self, name):
        return self.lib.has(name)

    def call(self, name, ins=(), outs=(), **kw):
        r = self.lib.call(name, ins, outs, **kw)
        for o in outs:
            if o is not None and getattr(o, "ndim", 0) == 3 and o.shape[2] == 4:
                with np.errstate(over="ignore"):
                    o[...] = o.astype(np.float16).astype(np.float32)
        return r


_cache = {}


def ref_lib():
    """The reference compiled for the CPU (oracle/_ref). Returns None if it has not been built / did not travel."""
    p = os.path.join(HERE, "_ref", "libmifx_ref.so")
    if not os.path.exists(p):
        return None
    if p not in _cache:
        _cache[p] = _Lib(p)
    return _cache[p]


def oracle_lib():
    """The hand-written CPU restatement (oracle/mifx_oracle.cpp); built on demand with g++."""
    p = os.path.join(HERE, "libmifx_oracle.so")
    if not os.path.exists(p):
        import importlib.util

        spec = importlib.util.spec_from_file_location("_oracle_build", os.path.join(HERE, "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build_oracle()
    if p not in _cache:
        _cache[p] = _Lib(p)
    return _cache[p]
