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


def store_unorm8(a):
    """What an R8_UNORM render target keeps of a float: clamp to [0, 1] (NaN -> 0), scale by 255, add 0.5, truncate; read back as code / 255 (all in fp32)."""
    x = np.where(np.isnan(a), np.float32(0), np.clip(a, np.float32(0), np.float32(1))).astype(np.float32)
    code = np.floor(x * np.float32(255.0) + np.float32(0.5)).astype(np.float32)
    return (code / np.float32(255.0)).astype(np.float32)


def store_unorm16(a):
    """R16_UNORM (the dilated circle of confusion of depth of field), as store_unorm8."""
    x = np.where(np.isnan(a), np.float32(0), np.clip(a, np.float32(0), np.float32(1))).astype(np.float32)
    code = np.floor(x * np.float32(65535.0) + np.float32(0.5)).astype(np.float32)
    return (code / np.float32(65535.0)).astype(np.float32)


def store_f16(a):
    with np.errstate(over="ignore"):
        return a.astype(np.float16).astype(np.float32)


def store_r11g11b10(a, alpha_reads_as=None):
    """(H, W, 4) colour through an R11G11B10_FLOAT target (oracle/format_ref.py: unsigned small floats, round to nearest even, negative -> 0); the format has no
    alpha channel -- alpha_reads_as=1.0 is what a sampler returns, None leaves the channel alone (it is never read)."""
    import format_ref as F

    out = a.copy()
    for c, m in ((0, 6), (1, 6), (2, 5)):
        out[..., c] = F.ufloat_to_float(F.float_to_ufloat(a[..., c].astype(np.float32), m), m)
    if alpha_reads_as is not None:
        out[..., 3] = np.float32(alpha_reads_as)
    return out


class QuantizingLib:
    """TEST INFRASTRUCTURE: a checker library whose passes store their images into the reference's own target formats -- the format-emulation mode of SURVEY.md
    section 0.2.  After a pass has run, each output is replaced by what its target keeps:
        every (H, W, 4) colour image        RGBA16_FLOAT (round to nearest even binary16) unless the table below names another format
        the planes named in STORES          R8_UNORM / R16_FLOAT / RG16_FLOAT / R11G11B10_FLOAT as the reference allocates them
                                            (ScreenSpaceAmbientOcclusion.hpp:255-256, ScreenSpaceReflection.cpp:155-290, PostFXContext.cpp:281, Bloom.cpp:111-137,
                                            DepthOfField.cpp:196-289)
    Used against the native-storage build of the product library (libmifx_h4.so); cube maps are produced with the plain library."""

    # pass (name without the ref_ / oracle_ prefix; a trailing * matches the permutations) -> format of each output in order (None: full precision)
    STORES = [
        ("closest_motion", ["rg16f"]),
        ("ssao_compute_ao_*", ["unorm8"]),
        ("ssao_bilateral_upsampling", ["unorm8"]),
        ("ssao_temporal_accumulation", ["unorm8", "r16f"]),
        ("ssao_convoluted_history_mip", ["unorm8", None]),
        ("ssao_resampled_history", ["unorm8"]),
        ("ssao_spatial_reconstruction", ["unorm8"]),
        ("ssr_mask_roughness", ["unorm8", None]),
        ("ssr_spatial_reconstruction*", [None, "r16f", "r16f"]),
        ("ssr_temporal_accumulation", [None, "r16f"]),
        ("bloom_prefilter", ["r11g11b10"]),
        ("bloom_downsample", ["r11g11b10"]),
        ("bloom_upsample", ["r11g11b10"]),
        # depth of field (DepthOfField.cpp:196-289): CoC and its history R16_FLOAT; separated / dilated / blurred CoC R16_UNORM; the combined output R11G11B10_FLOAT
        ("dof_coc", ["r16f"]),
        ("dof_temporal_coc", ["r16f"]),
        ("dof_separated_coc", ["unorm16"]),
        ("dof_dilation_coc", ["unorm16"]),
        ("dof_blur*", ["unorm16"]),
        ("dof_combine", ["r11g11b10_a1"]),
    ]

    def __init__(self, lib):
        self.lib = lib
        self.path = lib.path
        # FEATURE_FLAG_HALF_PRECISION_DEPTH of PostFXContext / ScreenSpaceAmbientOcclusion: the reprojected + previous depth (PostFXContext.cpp:259-270) and SSAO's two depth
        # pyramids (ScreenSpaceAmbientOcclusion.cpp:95-97) are R16_UNORM targets; the test that sets the flag on the product sets these
        self.depth16 = {"postfx": False, "ssao": False}

    def store_depth16(self, kind, a):
        """What a copy of a depth plane into one of those targets keeps (CopyTextureDepth / ComputePreviousDepth): oracle/cpu_chain.py calls it where the reference copies."""
        return store_unorm16(a) if self.depth16[kind] else a.copy()

    def has(self, name):
        return self.lib.has(name)

    def _formats(self, name):
        base = name.split("_", 1)[1] if name.startswith(("ref_", "oracle_")) else name
        if self.depth16["postfx"] and base == "reprojected_depth":
            return ["unorm16"]
        if self.depth16["ssao"] and base == "ssao_prefiltered_depth_mip":
            return ["unorm16"]
        for key, fmts in self.STORES:
            if (key.endswith("*") and base.startswith(key[:-1])) or base == key:
                return ["unorm8", "unorm16"] if self.depth16["ssao"] and base == "ssao_convoluted_history_mip" else fmts
        return []

    def call(self, name, ins=(), outs=(), **kw):
        r = self.lib.call(name, ins, outs, **kw)
        fmts = self._formats(name)
        final_bloom = name.endswith("bloom_upsample") and list(kw.get("ival", [0]))[:1] == [3]  # the pass that writes Bloom's output target
        for i, o in enumerate(outs):
            if o is None:
                continue
            f = fmts[i] if i < len(fmts) else None
            if f == "unorm8":
                o[...] = store_unorm8(o)
            elif f in ("r16f", "rg16f"):
                o[...] = store_f16(o)
            elif f == "unorm16":
                o[...] = store_unorm16(o)
            elif f == "r11g11b10_a1":
                o[...] = store_r11g11b10(o, alpha_reads_as=1.0)
            elif f == "r11g11b10":
                o[...] = store_r11g11b10(o, alpha_reads_as=1.0 if final_bloom else None)
            elif getattr(o, "ndim", 0) == 3 and o.shape[2] == 4:
                o[...] = store_f16(o)
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
