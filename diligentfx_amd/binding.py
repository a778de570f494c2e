"""ctypes binding of libmifx.so (include/mifx.h). Device memory comes from torch tensors (plumbing only):
an image is a contiguous float32 CUDA tensor of shape (H, W) / (H, W, 2) / (H, W, 4).

There is NO CPU fallback here: if libmifx.so is missing, importing this module raises."""
import ctypes
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MIFX_STORAGE=h4 selects the RGBA16_FLOAT storage build of the library for the whole process (include/mifx.h: mifx_storage_mode); MIFX_LIB_PATH: an
# alternative build of the same library (the sanitizer build of the host objects)
STORAGE_H4 = os.environ.get("MIFX_STORAGE", "fp32").lower() in ("h4", "rgba16f", "f16")
LIB_PATH = os.environ.get("MIFX_LIB_PATH") or os.path.join(HERE, "libmifx_h4.so" if STORAGE_H4 else "libmifx.so")

MIFX_OK = 0
FORMAT_F32, FORMAT_F32X2, FORMAT_F32X4, FORMAT_F16X4 = 1, 2, 4, 8
FORMAT_U8, FORMAT_F16, FORMAT_F16X2, FORMAT_R11G11B10, FORMAT_U16 = 16, 32, 64, 128, 256  # the narrow planes of the native-storage build (include/mifx.h)

c_f = ctypes.c_float
c_i = ctypes.c_int32
c_u = ctypes.c_uint32
c_p = ctypes.c_void_p


class Image2D(ctypes.Structure):
    _fields_ = [("data", c_p), ("width", c_u), ("height", c_u), ("pitch_bytes", c_u), ("format", c_u)]


class Cubemap(ctypes.Structure):
    _fields_ = [("mip_data", c_p * 16), ("size", c_u), ("mip_count", c_u)]


class CameraAttribs(ctypes.Structure):
    _fields_ = [
        ("f4Position", c_f * 4), ("f4ViewportSize", c_f * 4),
        ("fNearPlaneZ", c_f), ("fFarPlaneZ", c_f), ("fNearPlaneDepth", c_f), ("fFarPlaneDepth", c_f),
        ("fSceneNearZ", c_f), ("fSceneFarZ", c_f), ("fSceneNearDepth", c_f), ("fSceneFarDepth", c_f),
        ("fHandness", c_f), ("uiFrameIndex", c_u), ("Padding0", c_f), ("Padding1", c_f),
        ("fFocusDistance", c_f), ("fFStop", c_f), ("fFocalLength", c_f), ("fSensorWidth", c_f),
        ("fSensorHeight", c_f), ("fExposure", c_f), ("f2Jitter", c_f * 2),
        ("mView", c_f * 16), ("mProj", c_f * 16), ("mViewProj", c_f * 16),
        ("mViewInv", c_f * 16), ("mProjInv", c_f * 16), ("mViewProjInv", c_f * 16),
        ("f4ExtraData", (c_f * 4) * 5),
    ]


class ToneMappingAttribs(ctypes.Structure):
    _fields_ = [
        ("iToneMappingMode", c_i), ("bAutoExposure", c_i), ("fMiddleGray", c_f), ("bLightAdaptation", c_i),
        ("fWhitePoint", c_f), ("fLuminanceSaturation", c_f), ("Padding0", c_u), ("Padding1", c_u),
        ("AgXSaturation", c_f), ("AgXSlope", c_f), ("AgXPower", c_f), ("AgXOffset", c_f),
    ]

    @classmethod
    def default(cls, mode=4):
        return cls(mode, 1, 0.18, 1, 3.0, 1.0, 0, 0, 1.0, 1.0, 1.0, 0.0)


class SSAOAttribs(ctypes.Structure):
    _fields_ = [
        ("EffectRadius", c_f), ("EffectFalloffRange", c_f), ("RadiusMultiplier", c_f), ("DepthMIPSamplingOffset", c_f),
        ("TemporalStabilityFactor", c_f), ("SpatialReconstructionRadius", c_f), ("ResetAccumulation", c_i), ("AlphaInterpolation", c_f),
        ("BitmaskThickness", c_f), ("Algorithm", c_u), ("Padding0", c_f), ("Padding1", c_f),
    ]

    @classmethod
    def default(cls):
        return cls(1.0, 0.615, 1.457, 3.3, 0.9, 4.0, 0, 1.0, 0.5, 0, 0.0, 0.0)


class SSRAttribs(ctypes.Structure):
    _fields_ = [
        ("DepthBufferThickness", c_f), ("RoughnessThreshold", c_f), ("MostDetailedMip", c_u), ("IsRoughnessPerceptual", c_i),
        ("RoughnessChannel", c_u), ("MaxTraversalIntersections", c_u), ("GGXImportanceSampleBias", c_f), ("SpatialReconstructionRadius", c_f),
        ("TemporalRadianceStabilityFactor", c_f), ("TemporalVarianceStabilityFactor", c_f), ("BilateralCleanupSpatialSigmaFactor", c_f),
        ("AlphaInterpolation", c_f),
    ]

    @classmethod
    def default(cls):
        return cls(0.025, 0.2, 0, 1, 0, 128, 0.3, 4.0, 1.0, 0.9, 0.9, 1.0)


class BloomAttribs(ctypes.Structure):
    _fields_ = [("Intensity", c_f), ("Threshold", c_f), ("SoftTreshold", c_f), ("Radius", c_f), ("AlphaInterpolation", c_f),
                ("Padding0", c_f), ("Padding1", c_f), ("Padding2", c_f)]

    @classmethod
    def default(cls):
        return cls(0.15, 1.0, 0.125, 0.75, 1.0, 0.0, 0.0, 0.0)


class DOFAttribs(ctypes.Structure):
    """DepthOfFieldAttribs -- Shaders/PostProcess/DepthOfField/public/DepthOfFieldStructures.fxh:31-56"""
    _fields_ = [("MaxCircleOfConfusion", c_f), ("TemporalStabilityFactor", c_f), ("BokehKernelRingCount", c_i), ("BokehKernelRingDensity", c_i),
                ("AlphaInterpolation", c_f), ("Padding0", c_f), ("Padding1", c_f), ("Padding2", c_f)]

    @classmethod
    def default(cls):
        return cls(0.01, 0.9375, 5, 7, 1.0, 0.0, 0.0, 0.0)


class TAAAttribs(ctypes.Structure):
    _fields_ = [("TemporalStabilityFactor", c_f), ("ResetAccumulation", c_i), ("SkipRejection", c_i), ("Padding0", c_f)]

    @classmethod
    def default(cls):
        return cls(0.9375, 0, 0, 0.0)


class PBRLightAttribs(ctypes.Structure):
    _fields_ = [
        ("Type", c_i), ("PosX", c_f), ("PosY", c_f), ("PosZ", c_f),
        ("DirectionX", c_f), ("DirectionY", c_f), ("DirectionZ", c_f), ("ShadowMapIndex", c_i),
        ("IntensityR", c_f), ("IntensityG", c_f), ("IntensityB", c_f), ("Range4", c_f),
        ("SpotAngleScale", c_f), ("SpotAngleOffset", c_f), ("Padding0", c_f), ("Padding1", c_f),
    ]


PBR_MAX_LIGHTS = 16


class PBRShadeAttribs(ctypes.Structure):
    _fields_ = [("IBLScale", c_f * 4), ("OcclusionStrength", c_f), ("EmissionScale", c_f), ("PrefilteredCubeLastMip", c_f),
                ("LightCount", c_i), ("Lights", PBRLightAttribs * PBR_MAX_LIGHTS), ("Workflow", c_i), ("Padding", c_i * 3)]


class PBRRendererShaderParameters(ctypes.Structure):  # PBR_Structures.fxh:126-149 (144 bytes)
    _fields_ = [("AverageLogLum", c_f), ("MiddleGray", c_f), ("WhitePoint", c_f), ("PrefilteredCubeLastMip", c_f), ("IBLScale", c_f * 4), ("OcclusionStrength", c_f),
                ("EmissionScale", c_f), ("PointSize", c_f), ("MipBias", c_f), ("LightCount", c_i), ("Time", c_f), ("DebugView", c_i), ("Padding0", c_f),
                ("UnshadedColor", c_f * 4), ("HighlightColor", c_f * 4), ("LoadingAnimation", c_f * 12)]


class PBRMaterialBasicAttribs(ctypes.Structure):  # PBR_Structures.fxh:154-180 (96 bytes)
    _fields_ = [("BaseColorFactor", c_f * 4), ("EmissiveFactor", c_f * 3), ("NormalScale", c_f), ("SpecularFactor", c_f * 3), ("ClearcoatNormalScale", c_f), ("Workflow", c_i),
                ("AlphaMode", c_i), ("AlphaMaskCutoff", c_f), ("MetallicFactor", c_f), ("RoughnessFactor", c_f), ("OcclusionFactor", c_f), ("ClearcoatFactor", c_f),
                ("ClearcoatRoughnessFactor", c_f), ("CustomData", c_f * 4)]


def pbr_frame_attribs(camera, prev_camera, renderer, lights, max_lights, shadow_maps=(), max_shadow_maps=0) -> bytes:
    """PBRFrameAttribs (RenderPBR_Structures.fxh:11-24) as the bytes of the renderer's frame constant buffer: Camera | PrevCamera | Renderer | Lights[max_lights] |
    ShadowMaps[max_shadow_maps]."""
    b = bytes(camera) + bytes(prev_camera) + bytes(renderer)
    b += b"".join(bytes(l) for l in lights) + bytes(64 * (max_lights - len(lights)))
    b += b"".join(bytes(s) for s in shadow_maps) + bytes(96 * (max_shadow_maps - len(shadow_maps)))
    return b


class DeviceDesc(ctypes.Structure):
    _fields_ = [("device", c_i), ("hip_stream", c_p)]


class PostFXCreateInfo(ctypes.Structure):
    _fields_ = [("sobol_256d", c_p), ("scrambling_tile", c_p)]


class FrameDesc(ctypes.Structure):
    _fields_ = [("Index", c_u), ("Width", c_u), ("Height", c_u), ("OutputWidth", c_u), ("OutputHeight", c_u)]


PImage = ctypes.POINTER(Image2D)


class PostFXRenderAttribs(ctypes.Structure):
    _fields_ = [("curr_depth", PImage), ("prev_depth", PImage), ("motion", PImage),
                ("curr_camera", ctypes.POINTER(CameraAttribs)), ("prev_camera", ctypes.POINTER(CameraAttribs))]


class SSAORenderAttribs(ctypes.Structure):
    _fields_ = [("postfx", c_p), ("depth", PImage), ("normal", PImage), ("attribs", ctypes.POINTER(SSAOAttribs))]


class SSRRenderAttribs(ctypes.Structure):
    _fields_ = [("postfx", c_p), ("color", PImage), ("depth", PImage), ("normal", PImage), ("material", PImage), ("motion", PImage),
                ("attribs", ctypes.POINTER(SSRAttribs))]


class TAARenderAttribs(ctypes.Structure):
    _fields_ = [("postfx", c_p), ("color", PImage), ("attribs", ctypes.POINTER(TAAAttribs))]


class BloomRenderAttribs(ctypes.Structure):
    _fields_ = [("postfx", c_p), ("color", PImage), ("attribs", ctypes.POINTER(BloomAttribs))]


class DOFRenderAttribs(ctypes.Structure):
    _fields_ = [("postfx", c_p), ("color", PImage), ("depth", PImage), ("attribs", ctypes.POINTER(DOFAttribs))]


class GBuffer(ctypes.Structure):
    _fields_ = [("base_color", PImage), ("normal", PImage), ("material", PImage), ("depth", PImage), ("emissive", PImage), ("occlusion", PImage)]


class IBL(ctypes.Structure):
    _fields_ = [("brdf_lut", PImage), ("irradiance", ctypes.POINTER(Cubemap)), ("prefiltered", ctypes.POINTER(Cubemap))]


PBR_LAYER_CLEAR_COAT, PBR_LAYER_SHEEN, PBR_LAYER_ANISOTROPY, PBR_LAYER_IRIDESCENCE, PBR_LAYER_TRANSMISSION = 1, 2, 4, 8, 16
PBR_LAYER_PLANES = ("clearcoat", "clearcoat_normal", "sheen", "anisotropy", "tangent", "iridescence", "transmission", "sheen_albedo_scaling_lut", "preintegrated_charlie")


class PBRLayers(ctypes.Structure):  # mifx_pbr_layers
    _fields_ = [("flags", c_u), ("iridescence_ior", c_f), ("anisotropy_rotation", c_f), ("padding", c_u)] + [(n, PImage) for n in PBR_LAYER_PLANES]


class NativeImage(ctypes.Structure):  # mifx_native_image
    _fields_ = [("data", c_p), ("width", c_u), ("height", c_u), ("pitch_bytes", c_u), ("format", c_u)]


class GBufferNative(ctypes.Structure):  # mifx_gbuffer_native
    _P = ctypes.POINTER(NativeImage)
    _fields_ = [("base_color", _P), ("normal", _P), ("material", _P), ("depth", _P), ("emissive", _P), ("occlusion", _P)]


NATIVE_FORMATS = {name: i + 1 for i, name in enumerate(
    ["R32_FLOAT", "RG32_FLOAT", "RGBA32_FLOAT", "R16_FLOAT", "RG16_FLOAT", "RGBA16_FLOAT", "R8_UNORM", "RG8_UNORM", "RGBA8_UNORM", "RGBA8_UNORM_SRGB",
     "R16_UNORM", "RG16_UNORM", "RGBA16_UNORM", "R11G11B10_FLOAT"])}


class PBRShadowMapInfo(ctypes.Structure):
    """PBRShadowMapInfo -- Shaders/PBR/public/PBR_Structures.fxh:336-347 (96 bytes)"""
    _fields_ = [("WorldToLightProjSpace", c_f * 16), ("UVScale", c_f * 2), ("UVBias", c_f * 2), ("ShadowMapSlice", c_f), ("Padding0", c_f), ("Padding1", c_f), ("Padding2", c_f)]


class ShadowMapArray(ctypes.Structure):  # mifx_shadow_map_array
    _fields_ = [("data", c_p), ("width", c_u), ("height", c_u), ("slices", c_u), ("pitch_bytes", c_u), ("slice_pitch_bytes", ctypes.c_uint64)]


class PBRShadows(ctypes.Structure):  # mifx_pbr_shadows
    _fields_ = [("shadow_map", ctypes.POINTER(ShadowMapArray)), ("shadow_maps", ctypes.POINTER(PBRShadowMapInfo)), ("shadow_map_count", c_u), ("pcf_filter_size", c_u)]


class SphereMap(ctypes.Structure):  # mifx_spheremap
    _fields_ = [("mip_data", c_p * 16), ("width", c_u), ("height", c_u), ("mip_count", c_u)]


class EnvMapRenderAttribs(ctypes.Structure):
    """EnvMapRenderer::RenderAttribs -- Components/interface/EnvMapRenderer.hpp:97-118"""
    _fields_ = [("env_map", ctypes.POINTER(Cubemap)), ("average_log_lum", c_f), ("mip_level", c_f), ("alpha", c_f), ("options", c_u), ("scale", c_f * 3),
                ("sphere_map", ctypes.POINTER(SphereMap))]


class CompositeAttribs(ctypes.Structure):
    _fields_ = [("color", PImage), ("specular_ibl", PImage), ("ssr", PImage), ("ssao", PImage), ("normal", PImage), ("base_color", PImage),
                ("material", PImage), ("brdf_lut", PImage), ("camera", ctypes.POINTER(CameraAttribs)), ("ssr_scale", c_f), ("ssao_scale", c_f),
                ("tone_mapping", ctypes.POINTER(ToneMappingAttribs)), ("ave_log_lum", c_f)]


class ChainFrame(ctypes.Structure):
    _fields_ = [("frame", FrameDesc), ("gbuffer", GBuffer), ("motion", PImage), ("prev_depth", PImage),
                ("curr_camera", ctypes.POINTER(CameraAttribs)), ("prev_camera", ctypes.POINTER(CameraAttribs)),
                ("ibl", ctypes.POINTER(IBL)), ("pbr", ctypes.POINTER(PBRShadeAttribs)), ("ssao", ctypes.POINTER(SSAOAttribs)),
                ("ssr", ctypes.POINTER(SSRAttribs)), ("taa", ctypes.POINTER(TAAAttribs)), ("bloom", ctypes.POINTER(BloomAttribs)),
                ("tone_mapping", ctypes.POINTER(ToneMappingAttribs)), ("ave_log_lum", c_f), ("ssr_scale", c_f), ("ssao_scale", c_f),
                ("background", c_f * 4), ("taa_feature_flags", c_u), ("tonemap_flags", c_u)]


SIZEOF_NAMES = {
    "image2d": Image2D, "cubemap": Cubemap, "camera_attribs": CameraAttribs, "tone_mapping_attribs": ToneMappingAttribs,
    "ssao_attribs": SSAOAttribs, "ssr_attribs": SSRAttribs, "bloom_attribs": BloomAttribs, "dof_attribs": DOFAttribs, "taa_attribs": TAAAttribs,
    "pbr_light_attribs": PBRLightAttribs, "pbr_shade_attribs": PBRShadeAttribs, "frame_desc": FrameDesc, "chain_frame": ChainFrame,
    "composite_attribs": CompositeAttribs, "gbuffer": GBuffer, "ibl": IBL, "pbr_shadow_map_info": PBRShadowMapInfo,
}

_lib = None


def load():
    """Loads libmifx.so; raises (loudly) when it has not been built -- there is no fallback path."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `python diligentfx_amd/build.py` (hipcc, gfx950). "
                              "diligentfx_amd has no CPU or PyTorch fallback.")
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.mifx_status_string.restype = ctypes.c_char_p
        _lib.mifx_last_error.restype = ctypes.c_char_p
        _lib.mifx_sizeof.restype = c_u
        _lib.mifx_sizeof.argtypes = [ctypes.c_char_p]
        _lib.mifx_abi_version.restype = c_u
        _lib.mifx_storage_mode.restype = c_u
        assert bool(_lib.mifx_storage_mode()) == STORAGE_H4 or os.environ.get("MIFX_LIB_PATH"), "the loaded library is not the storage build MIFX_STORAGE asks for"
    return _lib


def device_code_sha16(path=None):
    """Identity of the DEVICE code of a library build: the first 16 hex digits of the SHA-256 of its `.hip_fatbin` section (the gfx950 code objects hipcc embeds; host-side
    changes -- api_*.cpp -- leave it alone).  What ties a committed counter file to the kernels a run times: tools/pmc_traffic.py stamps it into profiles/rNN_pmc_traffic.json,
    bench.py compares it with the library it has loaded (roofline.traffic_matches_build).  None when the file has no such section."""
    import hashlib
    import struct

    path = path or LIB_PATH
    with open(path, "rb") as f:
        data = f.read()
    if data[:4] != b"\x7fELF" or data[4] != 2:
        return None
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum, shstrndx = struct.unpack_from("<HHH", data, 0x3A)
    sec = lambda i: struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize)  # noqa: E731  name, type, flags, addr, offset, size, link, info, align, entsize
    stroff = sec(shstrndx)[4]
    for i in range(shnum):
        name_off, _, _, _, off, size = sec(i)[:6]
        end = data.index(b"\0", stroff + name_off)
        if data[stroff + name_off:end] == b".hip_fatbin":
            return hashlib.sha256(data[off:off + size]).hexdigest()[:16]
    return None


def storage_dtype():
    """torch dtype of a 4-channel image in this process: float32, or float16 with MIFX_STORAGE=h4 (1- and 2-channel images are float32 in both)."""
    import torch

    return torch.float16 if load().mifx_storage_mode() == 1 else torch.float32


def plane_dtype(kind):
    """torch dtype of an effect-owned plane of the loaded library: kind "colour" (4 channels), "ao" / "roughness" (R8_UNORM in the native-storage build),
    "history_len" / "variance" (R16_FLOAT), "motion" (closest motion, RG16_FLOAT); float32 throughout in the fp32 build."""
    import torch

    if storage_dtype() == torch.float32:
        return torch.float32
    return {"colour": torch.float16, "ao": torch.uint8, "roughness": torch.uint8, "history_len": torch.float16, "variance": torch.float16, "motion": torch.float16}[kind]


def to_storage(t):
    """A float32 (H, W, 4) tensor as the 4-channel image of this process' storage mode (rounded to nearest-even binary16 with MIFX_STORAGE=h4)."""
    return t.to(storage_dtype()).contiguous() if t.dim() == 3 and t.shape[2] == 4 else t


class ShardInfo(ctypes.Structure):  # mifx_shard_info
    _fields_ = [("band_begin", ctypes.c_int32), ("band_end", ctypes.c_int32), ("halo_taa", ctypes.c_int32), ("halo_ssr", ctypes.c_int32),
                ("halo_ssao", ctypes.c_int32), ("gather_level", ctypes.c_int32), ("own_begin", ctypes.c_int32), ("own_end", ctypes.c_int32),
                ("ae_begin", ctypes.c_int32), ("ae_end", ctypes.c_int32)]


class CommStats(ctypes.Structure):  # mifx_comm_stats
    _fields_ = [("rank", ctypes.c_int32), ("world", ctypes.c_int32), ("is_rccl", ctypes.c_int32), ("ranks_in_communicator", ctypes.c_int32),
                ("groups", ctypes.c_uint64), ("bytes_sent", ctypes.c_uint64), ("bytes_received", ctypes.c_uint64),
                ("timed_groups", ctypes.c_uint32), ("exchange_ms_total", ctypes.c_float), ("exchange_ms_max", ctypes.c_float)]


SIZEOF_NAMES.update({"shard_info": ShardInfo, "comm_stats": CommStats})


class MifxError(RuntimeError):
    def __init__(self, status, detail):
        super().__init__(f"{status}: {detail}")
        self.status = status


def check(status):
    if status < 0:
        lib = load()
        raise MifxError(lib.mifx_status_string(status).decode(), lib.mifx_last_error().decode())
    return status


def image(t) -> Image2D:
    """torch CUDA float32 tensor (H,W) / (H,W,2) / (H,W,4), or float16 (H,W,4) (MIFX_FORMAT_F16X4, the RGBA16_FLOAT storage build), row-contiguous -> mifx_image2d."""
    import torch

    assert isinstance(t, torch.Tensor) and t.dtype in (torch.float32, torch.float16, torch.uint8, torch.int32), "images are float32 tensors (native-storage build: also float16 / uint8 / packed int32)"
    if t.dim() == 2:
        c = 1
    else:
        assert t.dim() == 3 and t.shape[2] in (2, 4), t.shape
        c = t.shape[2]
    assert t.stride(-1) == 1 and (t.dim() == 2 or t.stride(1) == c), "texels must be contiguous"
    if t.dtype == torch.uint8:   # R8_UNORM (ambient occlusion, SSR roughness)
        assert c == 1
        return Image2D(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0), FORMAT_U8)
    if t.dtype == torch.int32:   # R11G11B10_FLOAT, one packed texel per element (Bloom levels)
        assert c == 1
        return Image2D(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0) * 4, FORMAT_R11G11B10)
    if t.dtype == torch.float16:
        return Image2D(t.data_ptr(), t.shape[1], t.shape[0], t.stride(0) * 2, {1: FORMAT_F16, 2: FORMAT_F16X2, 4: FORMAT_F16X4}[c])
    pitch = t.stride(0) * 4
    return Image2D(t.data_ptr(), t.shape[1], t.shape[0], pitch, {1: FORMAT_F32, 2: FORMAT_F32X2, 4: FORMAT_F32X4}[c])


def camera_from_bytes(b: bytes) -> CameraAttribs:
    assert len(b) == ctypes.sizeof(CameraAttribs)
    return CameraAttribs.from_buffer_copy(b)


def as_bytes(struct) -> bytes:
    return bytes(struct)
