"""oracle/refhost.py -- TEST INFRASTRUCTURE: the reference's HOST code, executed.

oracle/_ref/libmifx_refhost.so is PostFXContext / ScreenSpaceAmbientOcclusion / ScreenSpaceReflection / TemporalAntiAliasing / Bloom / DepthOfField compiled from the sources where
they lie under /root/reference/PostProcess against a recording DiligentCore stand-in (oracle/refhost/dg).  `RefHost.frame()` drives them through one frame in the order
of HnPostProcessTask and returns what they asked the device to do as a list of commands; `Replayer.run()` executes such a list on numpy planes: clears, copies and
buffer updates directly, every Draw by calling the pass of oracle/_ref (the reference's shader source compiled for the CPU) that the bound pixel shader names, with the
textures the host code bound to the shader's variables BY NAME.  Reference host code + reference shaders = the reference's frame: the pin of the pass order, clears,
ping-pong, reset rules and mip loops that oracle/cpu_chain.py restates by hand and csrc/api_*.cpp follow (tests/test_host_sequence_vs_ref.py).

Only tests/ may import this module."""
import base64
import ctypes
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
CHANNELS = {"R32_FLOAT": 1, "R16_FLOAT": 1, "R16_UNORM": 1, "R8_UNORM": 1, "R8_UINT": 1, "D16_UNORM": 1, "D32_FLOAT": 1, "RG8_UNORM": 2, "RG16_FLOAT": 2, "RGBA16_FLOAT": 4,
            "RGBA32_FLOAT": 4, "R11G11B10_FLOAT": 4, "RG32_FLOAT": 2}  # (R11G11B10 planes carry a fourth channel like every colour plane of the fp32-storage contract; it is never read)


class _Frame(ctypes.Structure):
    _fields_ = [("index", ctypes.c_uint), ("width", ctypes.c_uint), ("height", ctypes.c_uint), ("postfx_flags", ctypes.c_uint), ("ssao_flags", ctypes.c_uint),
                ("ssr_flags", ctypes.c_uint), ("taa_flags", ctypes.c_uint), ("bloom_flags", ctypes.c_uint), ("timer_elapsed", ctypes.c_float), ("curr_camera", ctypes.c_void_p),
                ("prev_camera", ctypes.c_void_p), ("ssao", ctypes.c_void_p), ("ssr", ctypes.c_void_p), ("taa", ctypes.c_void_p), ("bloom", ctypes.c_void_p), ("dof_flags", ctypes.c_uint), ("dof", ctypes.c_void_p)]


class _EnvMap(ctypes.Structure):
    _fields_ = [("width", ctypes.c_uint), ("height", ctypes.c_uint), ("options", ctypes.c_uint), ("cube", ctypes.c_uint), ("env_size", ctypes.c_uint), ("env_mips", ctypes.c_uint),
                ("average_log_lum", ctypes.c_float), ("mip_level", ctypes.c_float), ("alpha", ctypes.c_float), ("scale", ctypes.c_float * 3), ("tone_mapping", ctypes.c_void_p),
                ("cameras", ctypes.c_void_p)]


def envmap_render(width, height, options, tone_mapping, cameras, average_log_lum, mip_level, alpha, scale, cube=True, env_size=32, env_mips=6):
    """Components/src/EnvMapRenderer.cpp executed for one frame (Prepare + Render): the command list (see refhost.cpp: refhost_envmap_render)."""
    lib = ctypes.CDLL(lib_path())
    lib.refhost_envmap_render.restype = ctypes.c_char_p
    tm = ctypes.create_string_buffer(bytes(tone_mapping), len(bytes(tone_mapping)))
    cams = ctypes.create_string_buffer(bytes(cameras), len(bytes(cameras)))
    e = _EnvMap(width, height, options, 1 if cube else 0, env_size, env_mips, average_log_lum, mip_level, alpha, (ctypes.c_float * 3)(*scale), ctypes.cast(tm, ctypes.c_void_p),
                ctypes.cast(cams, ctypes.c_void_p))
    return json.loads(lib.refhost_envmap_render(ctypes.byref(e)).decode())


def lib_path():
    return os.path.join(HERE, "_ref", "libmifx_refhost.so")


def available():
    return os.path.exists(lib_path())


class RefHost:
    """One set of the reference's effect objects (they keep their own state across frames: histories, last frame index, pipelines)."""

    SSAO, SSR, TAA, BLOOM, DOF = 1, 2, 4, 8, 16

    def __init__(self, effects=15):
        self.lib = ctypes.CDLL(lib_path())
        self.lib.refhost_create.restype = ctypes.c_void_p
        self.lib.refhost_create.argtypes = [ctypes.c_uint]
        self.lib.refhost_destroy.argtypes = [ctypes.c_void_p]
        self.lib.refhost_frame_execute.restype = ctypes.c_char_p
        self.lib.refhost_frame_execute.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        self.lib.refhost_taa_jitter.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_float)]
        self.lib.refhost_sizeof.argtypes = [ctypes.c_char_p]
        self.h = self.lib.refhost_create(effects)

    def sizeof(self, name):
        return self.lib.refhost_sizeof(name.encode())

    def close(self):
        if self.h:
            self.lib.refhost_destroy(self.h)
            self.h = None

    def frame(self, index, width, height, cam, prev_cam, ssao=None, ssr=None, taa=None, bloom=None, postfx_flags=0, ssao_flags=0, ssr_flags=0, taa_flags=0, bloom_flags=0, timer=1.0,
              dof=None, dof_flags=0):
        """cam / prev_cam / ssao / ssr / taa / bloom / dof: bytes or ctypes structs of the reference's attribute blocks (None: the effect is not executed this frame)."""
        keep = []

        def ptr(v):
            if v is None:
                return None
            b = ctypes.create_string_buffer(bytes(v), len(bytes(v)))
            keep.append(b)
            return ctypes.cast(b, ctypes.c_void_p)

        f = _Frame(index, width, height, postfx_flags, ssao_flags, ssr_flags, taa_flags, bloom_flags, timer, ptr(cam), ptr(prev_cam), ptr(ssao), ptr(ssr), ptr(taa), ptr(bloom), dof_flags, ptr(dof))
        return json.loads(self.lib.refhost_frame_execute(self.h, ctypes.byref(f)).decode())

    def taa_jitter(self):
        out = (ctypes.c_float * 2)()
        self.lib.refhost_taa_jitter(self.h, out)
        return out[0], out[1]


# ---------------------------------------------------------------------------------------------------------------- the replay
# pixel-shader entry point -> how the pass of oracle/_ref that holds the same shader is called:
#   fn        name of the pass without the ref_ prefix ({algo}: gtao / hbao / vbao from the SSAO_ALGORITHM macro); suffixes are appended from the macros
#   inputs    shader variable bound to each input slot, in the slot order of the wrapper (oracle/ref/ref_*.cpp: `ref_bind(ns::<variable>.s, a, <slot>)` -- checked
#             against the wrapper sources by Replayer.check_table())
#   mask      input slot that takes the reflection mask of ScreenSpaceReflection (the reference keeps it in a depth buffer and lets the depth test apply it)
#   cams      how many CameraAttribs the pass reads from cbCameraAttribs (current, previous)
#   attribs   the constant buffer variable that holds the effect's attribute block
#   ival      what the pass takes as uInstID / frame index: "start_vertex/3" (Draw's StartVertexLocation / 3), or an int
PASSES = {
    "ComputeBlueNoiseTexturePS": dict(fn="blue_noise", inputs=["g_SobolBuffer", "g_ScramblingTileBuffer"], ival="start_vertex/3"),
    "ComputeReprojectedDepthPS": dict(fn="reprojected_depth", inputs=["g_TextureDepth"], cams=2),
    "ComputeClosestMotionPS": dict(fn="closest_motion", inputs=["g_TextureDepth", "g_TextureMotion"], rev="POSTFX_OPTION_INVERTED_DEPTH"),
    "ComputeDownsampledDepthPS": dict(fn="ssao_downsampled_depth", inputs=["g_TextureDepth"]),
    "ComputePrefilteredDepthBufferPS": dict(fn="ssao_prefiltered_depth_mip", inputs=["g_TextureLastMip"], cams=1, attribs="cbScreenSpaceAmbientOcclusionAttribs", ival="start_vertex/3"),
    "ComputeAmbientOcclusionPS": dict(fn="ssao_compute_ao_{algo}", inputs=["g_TexturePrefilteredDepth", "g_TextureNormal", "g_TextureBlueNoise"], cams=1, attribs="cbScreenSpaceAmbientOcclusionAttribs",
                                      rev="SSAO_OPTION_INVERTED_DEPTH", half="SSAO_OPTION_HALF_RESOLUTION", halfprec="SSAO_OPTION_HALF_PRECISION_DEPTH"),
    "ComputeBilateralUpsamplingPS": dict(fn="ssao_bilateral_upsampling", inputs=["g_TextureDepth", "g_TextureOcclusion"], cams=1, attribs="cbScreenSpaceAmbientOcclusionAttribs"),
    "ComputeConvolutedDepthHistoryPS": dict(fn="ssao_convoluted_history_mip", inputs=["g_TextureHistoryLastMip", "g_TextureDepthLastMip"], ival="start_vertex/3"),
    "ComputeResampledHistoryPS": dict(fn="ssao_resampled_history", inputs=["g_TextureOcclusion", "g_TextureDepth", "g_TextureHistory", "g_TextureNormal"], cams=1, rev="SSAO_OPTION_INVERTED_DEPTH"),
    "ComputeHierarchicalDepthBufferPS": dict(fn="ssr_hiz_mip", inputs=["g_TextureLastMip"], rev="SSR_OPTION_INVERTED_DEPTH", ival="start_vertex/3"),
    "ComputeStencilMaskAndExtractRoughnessPS": dict(fn="ssr_mask_roughness", inputs=["g_TextureMaterialParameters", "g_TextureDepth"], attribs="cbScreenSpaceReflectionAttribs", rev="SSR_OPTION_INVERTED_DEPTH",
                                                    writes_mask=1),
    "ComputeDownsampledStencilMaskPS": dict(fn="ssr_downsampled_mask", inputs=["g_TextureRoughness", "g_TextureDepth"], attribs="cbScreenSpaceReflectionAttribs", writes_mask=0),
    "ComputeIntersectionPS": dict(fn="ssr_intersection", inputs=["g_TextureRadiance", "g_TextureNormal", "g_TextureRoughness", "g_TextureBlueNoise", "g_TextureDepthHierarchy", None, "g_TextureMotion"],
                                  mask=5, cams=1, attribs="cbScreenSpaceReflectionAttribs", rev="SSR_OPTION_INVERTED_DEPTH", half="SSR_OPTION_HALF_RESOLUTION", prev="SSR_OPTION_PREVIOUS_FRAME"),
    "BilateralCleanupDummy": None,
}
# entry points that several effects share: told apart by the shader file
PASSES_BY_FILE = {
    ("SSAO_ComputeTemporalAccumulation.fx", "ComputeTemporalAccumulationPS"): dict(
        fn="ssao_temporal_accumulation", inputs=["g_TextureCurrOcclusion", "g_TexturePrevOcclusion", "g_TextureHistory", "g_TextureCurrDepth", "g_TexturePrevDepth", "g_TextureMotion"], cams=2,
        attribs="cbScreenSpaceAmbientOcclusionAttribs", rev="SSAO_OPTION_INVERTED_DEPTH"),
    ("SSAO_ComputeSpatialReconstruction.fx", "ComputeSpatialReconstructionPS"): dict(
        fn="ssao_spatial_reconstruction", inputs=["g_TextureOcclusion", "g_TextureHistory", "g_TextureDepth", "g_TextureNormal"], cams=1, attribs="cbScreenSpaceAmbientOcclusionAttribs",
        rev="SSAO_OPTION_INVERTED_DEPTH"),
    ("SSR_ComputeSpatialReconstruction.fx", "ComputeSpatialReconstructionPS"): dict(
        fn="ssr_spatial_reconstruction", inputs=["g_TextureRoughness", "g_TextureNormal", "g_TextureDepth", "g_TextureRayDirectionPDF", "g_TextureIntersectSpecular", None], mask=5, cams=1,
        attribs="cbScreenSpaceReflectionAttribs", half="SSR_OPTION_HALF_RESOLUTION"),
    ("SSR_ComputeTemporalAccumulation.fx", "ComputeTemporalAccumulationPS"): dict(
        fn="ssr_temporal_accumulation", inputs=["g_TextureMotion", "g_TextureHitDepth", "g_TextureCurrDepth", "g_TextureCurrRadiance", "g_TextureCurrVariance", "g_TexturePrevDepth", "g_TexturePrevRadiance",
                                                 "g_TexturePrevVariance", None], mask=8, cams=2, attribs="cbScreenSpaceReflectionAttribs"),
    ("SSR_ComputeBilateralCleanup.fx", "ComputeBilateralCleanupPS"): dict(
        fn="ssr_bilateral_cleanup", inputs=["g_TextureDepth", "g_TextureNormal", "g_TextureRoughness", "g_TextureRadiance", "g_TextureVariance", None], mask=5, cams=1, attribs="cbScreenSpaceReflectionAttribs",
        rev="SSR_OPTION_INVERTED_DEPTH"),
    ("TAA_ComputeTemporalAccumulation.fx", "ComputeTemporalAccumulationPS"): dict(
        fn="taa_flags{taa}", inputs=["g_TextureCurrColor", "g_TexturePrevColor", "g_TextureMotion", "g_TextureCurrDepth", "g_TexturePrevDepth"], cams=2, attribs="cbTemporalAntiAliasingAttribs"),
    ("Bloom_ComputePrefilteredTexture.fx", "ComputePrefilteredTexturePS"): dict(fn="bloom_prefilter", inputs=["g_TextureInput"], attribs="cbBloomAttribs"),
    ("Bloom_ComputeDownsampledTexture.fx", "ComputeDownsampledTexturePS"): dict(fn="bloom_downsample", inputs=["g_TextureInput"]),
    ("Bloom_ComputeUpsampledTexture.fx", "ComputeUpsampledTexturePS"): dict(fn="bloom_upsample", inputs=["g_TextureInput", "g_TextureDownsampled"], attribs="cbBloomAttribs", ival="start_vertex/3"),
    # depth of field: DepthOfField.cpp:380-790 creates these eleven techniques
    ("DOF_ComputeCircleOfConfusion.fx", "ComputeCircleOfConfusionPS"): dict(fn="dof_coc", inputs=["g_TextureDepth"], cams=1, attribs="cbDepthOfFieldAttribs"),
    ("DOF_ComputeTemporalCircleOfConfusion.fx", "ComputeTemporalCircleOfConfusionPS"): dict(fn="dof_temporal_coc", inputs=["g_TextureCurrCoC", "g_TexturePrevCoC", "g_TextureMotion"], cams=1,
                                                                                            attribs="cbDepthOfFieldAttribs"),
    ("DOF_ComputeSeparatedCircleOfConfusion.fx", "ComputeSeparatedCoCPS"): dict(fn="dof_separated_coc", inputs=["g_TextureCoC"]),
    ("DOF_ComputeDilationCircleOfConfusion.fx", "ComputeDilationCoCPS"): dict(fn="dof_dilation_coc", inputs=["g_TextureLastMip"]),
    ("DOF_ComputeBlurredCircleOfConfusion.fx", "ComputeBlurredCoCPS"): dict(fn="dof_blur_{blur}", inputs=["g_TextureCoC", "g_TextureGaussKernel"]),
    ("DOF_ComputePrefilteredTexture.fx", "ComputePrefilteredTexturePS"): dict(fn="dof_prefilter", inputs=["g_TextureColor", "g_TextureCoC", "g_TextureDilationCoC"], attribs="cbDepthOfFieldAttribs"),
    ("DOF_ComputeBokehFirstPass.fx", "ComputeBokehPS"): dict(fn="dof_bokeh_first", inputs=["g_TextureColorCoCNear", "g_TextureColorCoCFar", "g_TextureBokehKernel", "g_TextureRadiance"], cams=1,
                                                             attribs="cbDepthOfFieldAttribs", karis="DOF_OPTION_KARIS_INVERSE"),
    ("DOF_ComputeBokehSecondPass.fx", "ComputeBokehPS"): dict(fn="dof_bokeh_second", inputs=["g_TextureColorCoCNear", "g_TextureColorCoCFar", "g_TextureBokehKernel"], cams=1, attribs="cbDepthOfFieldAttribs"),
    ("DOF_ComputePostfilteredTexture.fx", "ComputePostfilteredTexturePS"): dict(fn="dof_postfilter", inputs=["g_TextureColorCoCNear", "g_TextureColorCoCFar"]),
    ("DOF_ComputeCombinedTexture.fx", "ComputeCombinedTexturePS"): dict(fn="dof_combine", inputs=["g_TextureColor", "g_TextureCoC", "g_TextureDoFNearPlane", "g_TextureDoFFarPlane"], cams=1,
                                                                        attribs="cbDepthOfFieldAttribs"),
}
del PASSES["BilateralCleanupDummy"]
ALGO = {"0": "gtao", "1": "hbao", "2": "vbao"}
CAMERA_BYTES = 576


class Replayer:
    """Executes command lists of RefHost on numpy planes with the passes of oracle/_ref (pyref.ref_lib())."""

    def __init__(self, ref):
        self.ref = ref
        self.tex = {}    # id -> {"name", "format", "planes": [np.ndarray per mip]}
        self.buf = {}    # id -> bytes
        self.outputs = {}
        self.passes = []  # (debug-group path, pass name) of every draw of the last run, in order
        self.notes = []

    # -- helpers
    def plane(self, view):
        t = self.tex[view["tex"]]
        return t["planes"][view["mip"]]

    def planes(self, view):
        t = self.tex[view["tex"]]
        return t["planes"][view["mip"]:view["mip"] + view["mips"]]

    def _pass_of(self, ps):
        p = PASSES_BY_FILE.get((ps["file"], ps["entry"])) or PASSES.get(ps["entry"])
        if p is None:
            raise KeyError(f"no oracle/_ref pass is registered for the pixel shader {ps['file']}:{ps['entry']}")
        return p

    def run(self, commands, inputs, app_composite=None):
        """inputs: name -> array for the caller-owned textures of the frame ("depth", "prev_depth", "motion", "normal", "material", "color"); app_composite(self) -> array:
        what the application draws into the frame between SSAO and TAA (default: the colour input)."""
        self.passes = []
        for c in commands:
            op = c["op"]
            if op == "note":
                self.notes.append(c["what"])
                continue
            if op == "error":
                raise RuntimeError("the reference host code reported: " + c["what"])
            if op == "create_texture":
                ch = CHANNELS[c["format"]]
                planes = []
                for m in range(c["mips"]):
                    w, h = max(c["w"] >> m, 1), max(c["h"] >> m, 1)
                    planes.append(np.zeros((h, w) if ch == 1 else (h, w, ch), np.float32))
                if "data_b64" in c:
                    if c["format"].endswith("_FLOAT"):  # (32-bit float formats only: the DOF kernel tables)
                        assert c["format"] in ("R32_FLOAT", "RG32_FLOAT", "RGBA32_FLOAT"), c["format"]
                        planes[0] = np.frombuffer(base64.b64decode(c["data_b64"]), np.float32).reshape(planes[0].shape).copy()
                    else:
                        raw = np.frombuffer(base64.b64decode(c["data_b64"]), np.uint8).astype(np.float32)
                        planes[0] = raw.reshape(c["h"], c["w"]).copy()
                self.tex[c["id"]] = {"name": c["name"], "format": c["format"], "planes": planes}
            elif op == "destroy_texture":
                self.tex.pop(c["id"], None)
            elif op in ("create_buffer", "update_buffer"):
                self.buf[c["id"] if op == "create_buffer" else c["buf"]] = base64.b64decode(c["bytes_b64"])
            elif op == "frame":
                self.frame_inputs = c["inputs"]
                for name, tid in c["inputs"].items():
                    if name in inputs:
                        a = np.ascontiguousarray(inputs[name], np.float32)
                        assert a.shape == self.tex[tid]["planes"][0].shape, (name, a.shape, self.tex[tid]["planes"][0].shape)
                        self.tex[tid]["planes"][0] = a.copy()
            elif op == "update_texture":
                t = self.tex[c["tex"]]
                assert t["format"] in ("R32_FLOAT", "RG32_FLOAT", "RGBA32_FLOAT") and c["slice"] == 0, c
                x0, x1, y0, y1 = c["box"]
                p = t["planes"][c["mip"]]
                p[y0:y1, x0:x1] = np.frombuffer(base64.b64decode(c["data_b64"]), np.float32).reshape(p[y0:y1, x0:x1].shape)
            elif op == "clear":
                p = self.plane(c["view"])
                col = np.asarray(c["color"], np.float32)
                p[...] = col[0] if p.ndim == 2 else col[:p.shape[2]]
            elif op == "clear_depth":
                self.plane(c["view"])[...] = np.float32(c["depth"])
            elif op == "copy":
                self.tex[c["dst"]]["planes"][c["dst_mip"]][...] = self.tex[c["src"]]["planes"][c["src_mip"]]
            elif op == "app_composite":
                src = app_composite(self) if app_composite is not None else self.tex[self.frame_inputs["color"]]["planes"][0]
                self.tex[c["dst"]]["planes"][0][...] = src
            elif op == "output":
                self.outputs[c["effect"]] = self.plane(c["view"])
            elif op == "draw":
                self._draw(c)
            else:
                raise KeyError(op)
        return self.outputs

    def _draw(self, c):
        ps = c["ps"]
        group = "/".join(c["groups"])
        if ps["entry"] == "main" and ps.get("name") == "CopyTexturePS":
            # PostFXContext's texture copy (a full-screen draw that samples g_Texture at the texel centres of a target of the same size)
            src = self.plane(c["vars"]["g_Texture"])
            dst = self.plane(c["rtvs"][0])
            assert src.shape == dst.shape, (c["pso"], src.shape, dst.shape)
            dst[...] = src
            self.passes.append((group, "copy:" + c["pso"]))
            return
        p = self._pass_of(ps)
        m = ps["macros"]
        on = lambda key: key in p and m.get(p[key], "0") not in ("0", "")  # noqa: E731
        name = p["fn"].format(algo=ALGO.get(m.get("SSAO_ALGORITHM", "0"), "?"), blur="xy"[int(m.get("DOF_CIRCLE_OF_CONFUSION_BLUR_TYPE", "0"))],
                              taa=(int(m.get("TAA_OPTION_GAUSSIAN_WEIGHTING", "0")) | int(m.get("TAA_OPTION_BICUBIC_FILTER", "0")) << 1 | int(m.get("TAA_OPTION_YCOCG_COLOR_SPACE", "0")) << 2))
        for key, suffix in (("prev", "_prev"), ("half", "_half"), ("halfprec", "_halfprec"), ("rev", "_rev"), ("karis", "_karis")):
            if on(key):
                name += suffix
        ins = []
        for slot, var in enumerate(p["inputs"]):
            if var is None:
                ins.append(None)
                continue
            v = c["vars"].get(var)
            if v is None:
                ins.append(None)  # (a variable the shader permutation does not read: g_TextureMotion of the ray march without PREVIOUS_FRAME)
                continue
            pl = self.planes(v)
            ins.append(pl if len(pl) > 1 else pl[0])
        depth = c.get("depth") or {}
        dsv = c.get("dsv")
        if "mask" in p:
            assert dsv is not None and depth.get("enable") and not depth.get("write") and depth.get("func") == "LESS", (c["pso"], depth, dsv)
            frag = np.float32(float(c["vs"]["macros"].get("TRIANGLE_DEPTH", "0.0")))
            ins[p["mask"]] = (frag < self.plane(dsv)).astype(np.float32)  # the depth test of a full-screen triangle at TRIANGLE_DEPTH against the mask buffer
        outs = [self.plane(r) for r in c["rtvs"]]
        if "writes_mask" in p:
            # ALWAYS + depth writes: every fragment that is not discarded stores the triangle's depth (1.0) into the mask buffer
            assert dsv is not None and depth.get("enable") and depth.get("write") and depth.get("func") == "ALWAYS", (c["pso"], depth)
            frag = np.float32(float(c["vs"]["macros"].get("TRIANGLE_DEPTH", "0.0")))
            assert frag == 1.0, frag
            mask = np.zeros_like(self.plane(dsv))
            outs = outs + [mask] if p["writes_mask"] == 1 else [mask]
        kw = {}
        cb = c["vars"].get("cbCameraAttribs")
        if p.get("cams"):
            cams = self.buf[cb["buf"]]
            kw["cam0"] = cams[:CAMERA_BYTES]
            if p["cams"] == 2:
                kw["cam1"] = cams[CAMERA_BYTES:2 * CAMERA_BYTES]
        if p.get("attribs"):
            kw["attribs"] = self.buf[c["vars"][p["attribs"]]["buf"]]
        iv = p.get("ival")
        if iv == "start_vertex/3":
            kw["ival"] = [c["start_vertex"] // 3]
        elif iv == "start_vertex":
            kw["ival"] = [c["start_vertex"]]
        assert c["instances"] == 1, c
        self.ref.call("ref_" + name, ins, outs, **kw)
        if "writes_mask" in p:
            d = self.plane(dsv)
            d[mask != 0] = np.float32(1.0)
        self.passes.append((group, name))

    # -- the table above against the wrapper sources
    @staticmethod
    def check_table():
        """Every (variable, slot) pair of PASSES / PASSES_BY_FILE equals a `ref_bind(<namespace>::<variable>.s, a, <slot>)` line of the wrapper that defines the pass."""
        src = {}
        d = os.path.join(HERE, "ref")
        for fn in os.listdir(d):
            if fn.endswith((".cpp", ".inc")):
                src[fn] = open(os.path.join(d, fn)).read()
        problems = []
        for key, p in list(PASSES.items()) + list(PASSES_BY_FILE.items()):
            base = p["fn"].format(algo="gtao", taa=0, blur="x")
            owner = None
            for fn, text in src.items():
                if re.search(r"\bref_" + re.escape(base) + r"\b", text) or (base.startswith("ssao_compute_ao") and "A3_ENTRY" in text and "ref_bind" in text) \
                        or (base == "ssr_intersection" and "R4FN" in text and "ref_bind" in text) or (base.startswith("taa_flags") and "T1_ENTRY" in text and "ref_bind" in text) \
                        or (base == "dof_blur_x" and "D5FN" in text and "ref_bind" in text) or (base == "dof_bokeh_first" and "D7FN" in text and "ref_bind" in text):
                    if "ref_bind" in text:
                        owner = fn
                        break
            if owner is None:
                problems.append(f"{key}: no wrapper defines ref_{base}")
                continue
            binds = {int(s): v for v, s in re.findall(r"ref_bind(?:_cube)?\(\w+::(\w+)\.s, a, (\d+)\)", src[owner])}
            for slot, var in enumerate(p["inputs"]):
                if var is not None and binds.get(slot) != var:
                    problems.append(f"{key}: slot {slot} is {binds.get(slot)} in {owner}, {var} in the table")
        return problems
