"""Thin Python host mirror of the reference's pass interfaces on top of the C ABI (include/mifx.h).

Names follow the reference classes (PostFXContext, ScreenSpaceAmbientOcclusion, ScreenSpaceReflection,
TemporalAntiAliasing, Bloom) and their PrepareResources / Execute / Get*SRV protocol; tensors are torch CUDA
float32 tensors used purely as device-memory handles.  Every call goes through libmifx.so."""
import ctypes

import torch

from . import binding as B


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _view(desc: B.Image2D, device):
    """Wraps an effect-owned plane (mifx_image2d) as a torch tensor view without copying (valid until the next prepare)."""
    # channels, bytes per element, element type.  The native-storage build hands out float16 / uint8 (R8_UNORM) / int32 (one packed R11G11B10_FLOAT texel) / int16 (the
    # bits of an R16_UNORM code) views:
    # widen() below turns any of them into float32 values.
    c, eb, ts = {B.FORMAT_F32: (1, 4, "<f4"), B.FORMAT_F32X2: (2, 4, "<f4"), B.FORMAT_F32X4: (4, 4, "<f4"), B.FORMAT_F16X4: (4, 2, "<f2"), B.FORMAT_F16: (1, 2, "<f2"),
                 B.FORMAT_F16X2: (2, 2, "<f2"), B.FORMAT_U8: (1, 1, "|u1"), B.FORMAT_R11G11B10: (1, 4, "<i4"), B.FORMAT_U16: (1, 2, "<i2")}[desc.format]
    pitch_f = desc.pitch_bytes // eb
    n = pitch_f * desc.height

    class _Holder:  # __cuda_array_interface__ provider
        pass

    h = _Holder()
    h.__cuda_array_interface__ = {"shape": (n,), "typestr": ts, "data": (desc.data, False), "version": 2}
    flat = torch.as_tensor(h, device=device)
    rows = flat.view(desc.height, pitch_f)[:, : desc.width * c]
    return rows.view(desc.height, desc.width) if c == 1 else rows.unflatten(1, (desc.width, c))


def widen(t):
    """float32 values of a plane view of any storage type (see _view): float16 -> float, uint8 -> / 255 (R8_UNORM), int32 -> the three unsigned small floats of
    an R11G11B10_FLOAT texel (H, W, 3), int16 -> / 65535 (the bits of an R16_UNORM code)."""
    if t.dtype == torch.uint8:   # (the quotient in float64, then rounded: the correctly rounded code / N whatever the device's float32 division does)
        return (t.double() / 255.0).float()
    if t.dtype == torch.int16:
        return ((t.int() & 0xFFFF).double() / 65535.0).float()
    if t.dtype == torch.int32:
        def ufloat(v, m):  # 5 exponent bits, m mantissa bits, no sign
            e, f = v >> m, (v & ((1 << m) - 1)).float()
            sub = f * (2.0 ** -(14 + m))
            nrm = torch.ldexp(1.0 + f * (2.0 ** -m), (e - 15).int())
            r = torch.where(e == 0, sub, nrm)
            return torch.where(e == 31, torch.where(f == 0, torch.full_like(r, float("inf")), torch.full_like(r, float("nan"))), r)
        v = t.long() & 0xFFFFFFFF
        return torch.stack([ufloat(v & 0x7FF, 6), ufloat((v >> 11) & 0x7FF, 6), ufloat(v >> 22, 5)], dim=-1)
    return t.float()


class PostFXContext:
    """== Diligent::PostFXContext (PostProcess/Common/interface/PostFXContext.hpp:48-263)."""

    def __init__(self, device=0, sobol_256d=None, scrambling_tile=None):
        self.lib = B.load()
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        dev = B.DeviceDesc(self.device.index or 0, _stream_ptr(self.device))
        info = B.PostFXCreateInfo()
        self._keep = []
        if sobol_256d is not None:
            s = bytes(bytearray(sobol_256d))
            t = bytes(bytearray(scrambling_tile))
            assert len(s) == 256 and len(t) == 128 * 128 * 8
            sb, tb = ctypes.create_string_buffer(s, len(s)), ctypes.create_string_buffer(t, len(t))
            self._keep += [sb, tb]
            info.sobol_256d = ctypes.cast(sb, ctypes.c_void_p)
            info.scrambling_tile = ctypes.cast(tb, ctypes.c_void_p)
        self.handle = ctypes.c_void_p()
        B.check(self.lib.mifx_postfx_create(ctypes.byref(dev), ctypes.byref(info), ctypes.byref(self.handle)))
        self.frame = None

    def close(self):
        if self.handle:
            self.lib.mifx_postfx_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def sync_stream(self):
        B.check(self.lib.mifx_postfx_set_stream(self.handle, _stream_ptr(self.device)))

    def set_static_ibl(self, enable=True):
        """mifx_postfx_set_static_ibl: the IBL maps handed to the shade do not change from call to call (their apron copy is then made once)."""
        B.check(self.lib.mifx_postfx_set_static_ibl(self.handle, ctypes.c_int32(1 if enable else 0)))

    FEATURE_FLAG_REVERSED_DEPTH, FEATURE_FLAG_HALF_PRECISION_DEPTH, FEATURE_FLAG_TEMPORAL_UPSCALING = 1, 2, 4  # PostFXContext.hpp:53-57

    def prepare_resources(self, index, width, height, feature_flags=0, output_width=None, output_height=None):
        self.frame = B.FrameDesc(index, width, height, output_width or width, output_height or height)
        self.sync_stream()
        B.check(self.lib.mifx_postfx_prepare(self.handle, ctypes.byref(self.frame), feature_flags))

    def execute(self, curr_depth, prev_depth, motion, curr_camera: B.CameraAttribs, prev_camera: B.CameraAttribs):
        imgs = [B.image(curr_depth), B.image(prev_depth), B.image(motion)]
        a = B.PostFXRenderAttribs(ctypes.pointer(imgs[0]), ctypes.pointer(imgs[1]), ctypes.pointer(imgs[2]), ctypes.pointer(curr_camera),
                                  ctypes.pointer(prev_camera))
        self._inputs = (curr_depth, prev_depth, motion)
        B.check(self.lib.mifx_postfx_execute(self.handle, ctypes.byref(a)))

    def _get(self, fn, *args):
        d = B.Image2D()
        B.check(fn(self.handle, *args, ctypes.byref(d)))
        return _view(d, self.device)

    def get_reprojected_depth(self):
        return self._get(self.lib.mifx_postfx_get_reprojected_depth)

    def get_closest_motion_vectors(self):
        return self._get(self.lib.mifx_postfx_get_closest_motion)

    def get_previous_depth(self):
        return self._get(self.lib.mifx_postfx_get_previous_depth)

    def get_2d_blue_noise(self, dimension):
        return self._get(self.lib.mifx_postfx_get_blue_noise, ctypes.c_int32(dimension))

    # -- PostFXContext's public texture helpers (PostFXContext.hpp:114, 168-172)
    def get_supported_features(self):
        """PostFXContext::GetSupportedFeatures as a dict of the four capability flags."""
        f = (ctypes.c_int32 * 4)()
        B.check(self.lib.mifx_postfx_get_supported_features(self.handle, f))
        return dict(zip(("TransitionSubresources", "TextureSubresourceViews", "CopyDepthToColor", "ShaderBaseVertexOffset"), (bool(v) for v in f)))

    def clear_render_target(self, target, clear_color):
        """PostFXContext::ClearRenderTarget: every texel of `target` := clear_color (4 floats, one per channel)."""
        self.sync_stream()
        i = B.image(target)
        B.check(self.lib.mifx_postfx_clear_render_target(self.handle, ctypes.byref(i), (ctypes.c_float * 4)(*clear_color)))

    def copy_texture_depth(self, src, dst):
        self.sync_stream()
        s, d = B.image(src), B.image(dst)
        B.check(self.lib.mifx_postfx_copy_texture_depth(self.handle, ctypes.byref(s), ctypes.byref(d)))

    def copy_texture_color(self, src, dst):
        self.sync_stream()
        s, d = B.image(src), B.image(dst)
        B.check(self.lib.mifx_postfx_copy_texture_color(self.handle, ctypes.byref(s), ctypes.byref(d)))

    # -- stand-alone full-screen passes recorded on this context's stream
    def tone_map(self, hdr, attribs: B.ToneMappingAttribs, ave_log_lum, flags=0, out=None):
        """Full-screen ToneMap() (ToneMapping.fxh:87-226), see mifx_tonemap_execute."""
        if out is None:
            out = torch.empty_like(hdr)
        i, o = B.image(hdr), B.image(out)
        B.check(self.lib.mifx_tonemap_execute(self.handle, ctypes.byref(i), ctypes.byref(o), ctypes.byref(attribs), ctypes.c_float(ave_log_lum),
                                              ctypes.c_uint32(flags)))
        return out

    def tone_map_native(self, hdr, attribs: B.ToneMappingAttribs, ave_log_lum, fmt: str, flags=0, pitch_bytes=None):
        """ToneMap() stored in the target's own format (mifx_tonemap_execute_native): torch.uint8 (H, pitch_bytes), bit-identical to
        image_export(tone_map(...), fmt) without the fp32 intermediate."""
        raw, dst = _native_target(self, hdr.shape[0], hdr.shape[1], fmt, pitch_bytes, hdr.device)
        i = B.image(hdr)
        B.check(self.lib.mifx_tonemap_execute_native(self.handle, ctypes.byref(i), ctypes.byref(dst), ctypes.byref(attribs), ctypes.c_float(ave_log_lum),
                                                     ctypes.c_uint32(flags)))
        return raw


def _native_target(ctx, h, w, fmt, pitch_bytes, device):
    ts = ctx.lib.mifx_native_format_texel_size(ctypes.c_uint32(B.NATIVE_FORMATS[fmt]))
    pitch = pitch_bytes or w * ts
    raw = torch.zeros((h, pitch), dtype=torch.uint8, device=device)
    return raw, B.NativeImage(raw.data_ptr(), w, h, pitch, B.NATIVE_FORMATS[fmt])


class IBLResources:
    """Device-side IBL inputs (mifx_ibl): BRDF LUT (H,W,2|4), irradiance cube and prefiltered cube as lists of (6*s, s, 4) mips."""

    def __init__(self, lut, irradiance_mips, prefiltered_mips):
        self.lut, self.irr, self.pre = lut, list(irradiance_mips), list(prefiltered_mips)
        self._lut_img = B.image(lut)
        self._irr, self._pre = B.Cubemap(), B.Cubemap()
        for cm, mips in ((self._irr, self.irr), (self._pre, self.pre)):
            cm.size, cm.mip_count = mips[0].shape[1], len(mips)
            for i, m in enumerate(mips):
                assert m.is_contiguous() and m.dtype == torch.float32 and m.shape == (6 * (cm.size >> i), cm.size >> i, 4), (i, m.shape)
                cm.mip_data[i] = m.data_ptr()
        self.struct = B.IBL(ctypes.pointer(self._lut_img), ctypes.pointer(self._irr), ctypes.pointer(self._pre))


def _cubemap(mips):
    cm = B.Cubemap()
    cm.size, cm.mip_count = mips[0].shape[1], len(mips)
    for i, m in enumerate(mips):
        assert m.is_contiguous() and m.dtype == torch.float32 and m.shape == (6 * (cm.size >> i), cm.size >> i, 4), (i, m.shape)
        cm.mip_data[i] = m.data_ptr()
    return cm


def _spheremap(mips):
    sm = B.SphereMap()
    sm.height, sm.width, sm.mip_count = mips[0].shape[0], mips[0].shape[1], len(mips)
    for i, m in enumerate(mips):
        assert m.is_contiguous() and m.dtype == torch.float32 and m.shape[2] == 4
        sm.mip_data[i] = m.data_ptr()
    return sm


def ibl_from_sphere_map(ctx: "PostFXContext", sphere_mips, irradiance_size=64, prefiltered_size=256, diffuse_samples=8192, specular_samples=256):
    """PrecomputeCubemaps from an equirectangular environment map (ENV_MAP_TYPE_SPHERE): returns (irradiance cube, [prefiltered cube mips]); mip 0 of the
    prefiltered cube (roughness 0) is the equirect -> cube conversion."""
    env = _spheremap(sphere_mips)
    dev = sphere_mips[0].device
    irr = torch.empty(6 * irradiance_size, irradiance_size, 4, device=dev)
    B.check(ctx.lib.mifx_ibl_compute_irradiance_map_sphere(ctx.handle, ctypes.byref(env), ctypes.c_void_p(irr.data_ptr()), ctypes.c_uint32(irradiance_size),
                                                           ctypes.c_uint32(diffuse_samples)))
    levels = prefiltered_size.bit_length()
    pre = []
    for m in range(levels):
        s = prefiltered_size >> m
        o = torch.empty(6 * s, s, 4, device=dev)
        B.check(ctx.lib.mifx_ibl_prefilter_env_map_sphere(ctx.handle, ctypes.byref(env), ctypes.c_void_p(o.data_ptr()), ctypes.c_uint32(s),
                                                          ctypes.c_float(m / max(levels - 1, 1)), ctypes.c_uint32(specular_samples)))
        pre.append(o)
    torch.cuda.synchronize(dev)
    return irr, pre


def cube_box_mips(cube):
    """(6*n, n, 4) -> full mip chain by 2x2 box filtering (how an application prepares the environment map SRV)."""
    n = cube.shape[1]
    mips = [cube.contiguous()]
    faces = cube.view(6, n, n, 4)
    while n > 1:
        faces = faces.view(6, n // 2, 2, n // 2, 2, 4).mean(dim=(2, 4))
        n //= 2
        mips.append(faces.reshape(6 * n, n, 4).contiguous())
    return mips


def image_import(ctx: "PostFXContext", raw, width, fmt: str, channels=4):
    """mifx_image_import: `raw` = torch.uint8 (H, pitch_bytes) holding a linear-layout image in the native format `fmt` (binding.NATIVE_FORMATS)
    -> float32 tensor (H, W[, channels])."""
    assert raw.dtype == torch.uint8 and raw.dim() == 2 and raw.stride(1) == 1
    h = raw.shape[0]
    out = torch.empty((h, width) if channels == 1 else (h, width, channels), dtype=torch.float32, device=raw.device)
    src = B.NativeImage(raw.data_ptr(), width, h, raw.stride(0), B.NATIVE_FORMATS[fmt])
    d = B.image(out)
    B.check(ctx.lib.mifx_image_import(ctx.handle, ctypes.byref(src), ctypes.byref(d)))
    return out


def image_export(ctx: "PostFXContext", img, fmt: str, pitch_bytes=None):
    """mifx_image_export: float32 tensor (H, W[, C]) -> torch.uint8 (H, pitch_bytes) in the native format `fmt`."""
    h, w = img.shape[0], img.shape[1]
    ts = ctx.lib.mifx_native_format_texel_size(ctypes.c_uint32(B.NATIVE_FORMATS[fmt]))
    pitch = pitch_bytes or w * ts
    raw = torch.zeros((h, pitch), dtype=torch.uint8, device=img.device)
    dst = B.NativeImage(raw.data_ptr(), w, h, pitch, B.NATIVE_FORMATS[fmt])
    s = B.image(img)
    B.check(ctx.lib.mifx_image_export(ctx.handle, ctypes.byref(s), ctypes.byref(dst)))
    return raw


ENVMAP_OPTION_FLAG_CONVERT_OUTPUT_TO_SRGB, ENVMAP_OPTION_FLAG_COMPUTE_MOTION_VECTORS = 1, 2


def render_env_map(ctx: "PostFXContext", env_mips, depth, color, motion, camera: B.CameraAttribs, prev_camera: B.CameraAttribs, tone_mapping: B.ToneMappingAttribs = None,
                   average_log_lum=0.3, mip_level=1.0, alpha=0.0, scale=(1.0, 1.0, 1.0), options=ENVMAP_OPTION_FLAG_COMPUTE_MOTION_VECTORS):
    """EnvMapRenderer::Prepare + Render (mifx_envmap_render): the environment colour (and motion vectors) on every pixel at the far-plane depth of
    `color` / `motion`, in place. Defaults = Hydrogent's call (HnRenderEnvMapTask.cpp:165-219): tone mapping NONE, mip 1, alpha 0, motion vectors."""
    sphere = env_mips[0].shape[0] != 6 * env_mips[0].shape[1]  # a cube mip is (6 n, n, 4); anything else is an equirectangular map
    env = _spheremap(env_mips) if sphere else _cubemap(env_mips)
    tm = tone_mapping if tone_mapping is not None else B.ToneMappingAttribs.default(0)
    a = B.EnvMapRenderAttribs(None if sphere else ctypes.pointer(env), average_log_lum, mip_level, alpha, options, (ctypes.c_float * 3)(*scale),
                              ctypes.pointer(env) if sphere else None)
    d, c = B.image(depth), B.image(color)
    m = B.image(motion) if motion is not None else None
    B.check(ctx.lib.mifx_envmap_render(ctx.handle, ctypes.byref(a), ctypes.byref(tm), ctypes.byref(camera), ctypes.byref(prev_camera), ctypes.byref(d), ctypes.byref(c),
                                       ctypes.byref(m) if m is not None else None))
    return color, motion


def precompute_ibl(ctx: "PostFXContext", env_cube, lut_size=512, irradiance_size=64, prefiltered_size=256, lut_samples=512, diffuse_samples=8192,
                   specular_samples=256):
    """PBR_Renderer::PrecomputeBRDF + PrecomputeCubemaps on the GPU (mifx_ibl_*); defaults are the reference's (PBR_Renderer.hpp:298,477-480)."""
    dev = ctx.device
    env_mips = cube_box_mips(env_cube)
    env = _cubemap(env_mips)
    ctx.sync_stream()
    lut = torch.empty(lut_size, lut_size, 2, device=dev)
    li = B.image(lut)
    B.check(ctx.lib.mifx_ibl_precompute_brdf_lut(ctx.handle, ctypes.byref(li), ctypes.c_uint32(lut_samples)))
    irr = torch.empty(6 * irradiance_size, irradiance_size, 4, device=dev)
    B.check(ctx.lib.mifx_ibl_compute_irradiance_map(ctx.handle, ctypes.byref(env), ctypes.c_void_p(irr.data_ptr()), ctypes.c_uint32(irradiance_size),
                                                    ctypes.c_uint32(diffuse_samples)))
    levels = prefiltered_size.bit_length()
    pre = []
    for m in range(levels):
        s = prefiltered_size >> m
        o = torch.empty(6 * s, s, 4, device=dev)
        B.check(ctx.lib.mifx_ibl_prefilter_env_map(ctx.handle, ctypes.byref(env), ctypes.c_void_p(o.data_ptr()), ctypes.c_uint32(s),
                                                   ctypes.c_float(m / max(levels - 1, 1)), ctypes.c_uint32(specular_samples)))
        pre.append(o)
    torch.cuda.synchronize(dev)
    return IBLResources(lut, [irr], pre)


def pbr_shade(ctx: "PostFXContext", gbuffer: dict, camera: B.CameraAttribs, attribs: B.PBRShadeAttribs, ibl: IBLResources, background=(0.0, 0.0, 0.0, 0.0),
              out_radiance=None, out_specular_ibl=None, want_specular_ibl=True, shadows=None):
    """PBR shading entry (mifx_pbr_shade_execute). gbuffer: dict with base_color, normal, material, depth [, emissive, occlusion].
    shadows: (shadow_map float32 tensor (slices, H, W), infos = sequence of 24-float PBRShadowMapInfo rows, pcf_filter_size) -> mifx_pbr_shade_execute_with_shadows."""
    ref = gbuffer["depth"]
    h, w = ref.shape
    if out_radiance is None:
        out_radiance = torch.empty(h, w, 4, device=ref.device, dtype=B.storage_dtype())
    if out_specular_ibl is None and want_specular_ibl:
        out_specular_ibl = torch.empty(h, w, 4, device=ref.device, dtype=B.storage_dtype())
    imgs = {k: B.image(gbuffer[k]) for k in ("base_color", "normal", "material", "depth", "emissive", "occlusion") if gbuffer.get(k) is not None}
    p = lambda k: ctypes.pointer(imgs[k]) if k in imgs else None  # noqa: E731
    g = B.GBuffer(p("base_color"), p("normal"), p("material"), p("depth"), p("emissive"), p("occlusion"))
    o0 = B.image(out_radiance)
    o1 = B.image(out_specular_ibl) if out_specular_ibl is not None else None
    bg = (ctypes.c_float * 4)(*background)
    ctx.sync_stream()
    if shadows is not None:
        sm, infos, pcf = shadows
        assert sm.dtype == torch.float32 and sm.dim() == 3 and sm.is_contiguous()
        arr = B.ShadowMapArray(sm.data_ptr(), sm.shape[2], sm.shape[1], sm.shape[0], sm.stride(1) * 4, sm.stride(0) * 4)
        rows = (B.PBRShadowMapInfo * len(infos))()
        for i, r in enumerate(infos):
            rows[i] = r if isinstance(r, B.PBRShadowMapInfo) else B.PBRShadowMapInfo.from_buffer_copy(r.astype("float32").tobytes())
        sh = B.PBRShadows(ctypes.pointer(arr), ctypes.cast(rows, ctypes.POINTER(B.PBRShadowMapInfo)), len(infos), pcf)
        B.check(ctx.lib.mifx_pbr_shade_execute_with_shadows(ctx.handle, ctypes.byref(g), ctypes.byref(camera), ctypes.byref(attribs), ctypes.byref(ibl.struct), ctypes.byref(sh), bg,
                                                            ctypes.byref(o0), ctypes.byref(o1) if o1 is not None else None))
        return out_radiance, out_specular_ibl
    B.check(ctx.lib.mifx_pbr_shade_execute(ctx.handle, ctypes.byref(g), ctypes.byref(camera), ctypes.byref(attribs), ctypes.byref(ibl.struct), bg, ctypes.byref(o0),
                                           ctypes.byref(o1) if o1 is not None else None))
    return out_radiance, out_specular_ibl


def pbr_shade_layers(ctx: "PostFXContext", gbuffer: dict, layers: dict, flags: int, camera: B.CameraAttribs, attribs: B.PBRShadeAttribs, ibl: IBLResources,
                     background=(0.0, 0.0, 0.0, 0.0), iridescence_ior=1.3, anisotropy_rotation=0.0, want_specular_ibl=True, shadows=None):
    """The shade with material layers (mifx_pbr_shade_execute_layers).  layers: dict of the planes of mifx_pbr_layers (binding.PBR_LAYER_PLANES) that the set `flags`
    (binding.PBR_LAYER_*) needs; optional planes may be missing.  shadows: as pbr_shade()."""
    ref = gbuffer["depth"]
    h, w = ref.shape
    out_radiance = torch.empty(h, w, 4, device=ref.device, dtype=B.storage_dtype())
    out_specular_ibl = torch.empty(h, w, 4, device=ref.device, dtype=B.storage_dtype()) if want_specular_ibl else None
    imgs = {k: B.image(gbuffer[k]) for k in ("base_color", "normal", "material", "depth", "emissive", "occlusion") if gbuffer.get(k) is not None}
    p = lambda k: ctypes.pointer(imgs[k]) if k in imgs else None  # noqa: E731
    g = B.GBuffer(p("base_color"), p("normal"), p("material"), p("depth"), p("emissive"), p("occlusion"))
    limgs = {k: B.image(layers[k]) for k in B.PBR_LAYER_PLANES if layers.get(k) is not None}
    ly = B.PBRLayers(flags, iridescence_ior, anisotropy_rotation, 0, *[ctypes.pointer(limgs[k]) if k in limgs else None for k in B.PBR_LAYER_PLANES])
    o0 = B.image(out_radiance)
    o1 = B.image(out_specular_ibl) if out_specular_ibl is not None else None
    bg = (ctypes.c_float * 4)(*background)
    ctx.sync_stream()
    sh = None
    if shadows is not None:
        sm, infos, pcf = shadows
        assert sm.dtype == torch.float32 and sm.dim() == 3 and sm.is_contiguous()
        arr = B.ShadowMapArray(sm.data_ptr(), sm.shape[2], sm.shape[1], sm.shape[0], sm.stride(1) * 4, sm.stride(0) * 4)
        rows = (B.PBRShadowMapInfo * len(infos))()
        for i, r in enumerate(infos):
            rows[i] = r if isinstance(r, B.PBRShadowMapInfo) else B.PBRShadowMapInfo.from_buffer_copy(r.astype("float32").tobytes())
        sh = B.PBRShadows(ctypes.pointer(arr), ctypes.cast(rows, ctypes.POINTER(B.PBRShadowMapInfo)), len(infos), pcf)
    B.check(ctx.lib.mifx_pbr_shade_execute_layers(ctx.handle, ctypes.byref(g), ctypes.byref(ly), ctypes.byref(camera), ctypes.byref(attribs), ctypes.byref(ibl.struct),
                                                  ctypes.byref(sh) if sh is not None else None, bg, ctypes.byref(o0), ctypes.byref(o1) if o1 is not None else None))
    return out_radiance, out_specular_ibl


HYDROGENT_GBUFFER_FORMATS = {"base_color": "RGBA8_UNORM", "normal": "RGBA16_FLOAT", "material": "RG8_UNORM", "depth": "R32_FLOAT", "emissive": "RGBA16_FLOAT",
                             "occlusion": "R8_UNORM"}  # HnBeginFrameTask.cpp:63-69 (emissive / occlusion are not Hydrogent targets: same classes of format)


def pbr_shade_native(ctx: "PostFXContext", gbuffer: dict, width, camera: B.CameraAttribs, attribs: B.PBRShadeAttribs, ibl: IBLResources, background=(0.0, 0.0, 0.0, 0.0),
                     out_format="RGBA16_FLOAT", want_specular_ibl=True):
    """mifx_pbr_shade_execute_native. gbuffer: name -> (torch.uint8 (H, pitch_bytes) tensor, format name); returns the radiance (and IBL) targets as raw
    torch.uint8 (H, pitch) tensors of `out_format`."""
    h = gbuffer["depth"][0].shape[0]
    imgs = {k: B.NativeImage(t.data_ptr(), width, h, t.stride(0), B.NATIVE_FORMATS[f]) for k, (t, f) in gbuffer.items() if t is not None}
    p = lambda k: ctypes.pointer(imgs[k]) if k in imgs else None  # noqa: E731
    g = B.GBufferNative(p("base_color"), p("normal"), p("material"), p("depth"), p("emissive"), p("occlusion"))
    dev = gbuffer["depth"][0].device
    raw0, o0 = _native_target(ctx, h, width, out_format, None, dev)
    raw1, o1 = _native_target(ctx, h, width, out_format, None, dev) if want_specular_ibl else (None, None)
    bg = (ctypes.c_float * 4)(*background)
    ctx.sync_stream()
    B.check(ctx.lib.mifx_pbr_shade_execute_native(ctx.handle, ctypes.byref(g), ctypes.byref(camera), ctypes.byref(attribs), ctypes.byref(ibl.struct), bg, ctypes.byref(o0),
                                                  ctypes.byref(o1) if o1 is not None else None))
    return raw0, raw1


def composite(ctx: "PostFXContext", color, specular_ibl, ssr, ssao, normal, base_color, material, lut, camera, ssr_scale=1.0, ssao_scale=1.0,
              tone_mapping=None, ave_log_lum=0.3, out=None):
    """SSR / SSAO composite (mifx_composite_execute), Hydrogent/shaders/HnPostProcess.psh:145-185."""
    if out is None:
        out = torch.empty_like(color)
    imgs = [B.image(t) for t in (color, specular_ibl, ssr, ssao, normal, base_color, material, lut)]
    a = B.CompositeAttribs(*[ctypes.pointer(i) for i in imgs], ctypes.pointer(camera), ssr_scale, ssao_scale,
                           ctypes.pointer(tone_mapping) if tone_mapping is not None else None, ave_log_lum)
    o = B.image(out)
    ctx.sync_stream()
    B.check(ctx.lib.mifx_composite_execute(ctx.handle, ctypes.byref(a), ctypes.byref(o)))
    return out


def _export_history(fx, channel_shapes):
    """mifx_<effect>_export_history into fresh tensors of the prepared size; returns (*planes, frame_index)."""
    # (a plane of the prepared size; not the SSR output: inside a chain that plane may be deferred, mifx_ssr_run_deferred_cleanup)
    ref = fx.get_intermediate("hist_radiance") if fx._prefix == "ssr" else (fx._output() if fx._prefix != "taa" else fx._output(ctypes.c_int32(0)))
    h, w = ref.shape[0], ref.shape[1]
    planes = [torch.empty((h, w) + tuple(c), device=fx.ctx.device, dtype=B.plane_dtype(kind)) for c, kind in channel_shapes]
    imgs = [B.image(p) for p in planes]
    idx = ctypes.c_uint32(0)
    B.check(getattr(fx.lib, f"mifx_{fx._prefix}_export_history")(fx.handle, *[ctypes.byref(i) for i in imgs], ctypes.byref(idx)))
    return (*planes, idx.value)


def _import_history(fx, planes, frame_index):
    imgs = [B.image(p) for p in planes]
    B.check(getattr(fx.lib, f"mifx_{fx._prefix}_import_history")(fx.handle, *[ctypes.byref(i) for i in imgs], ctypes.c_uint32(frame_index)))


class _Effect:
    """Common PrepareResources / Execute / Get*SRV plumbing of the effect objects."""

    _prefix = None

    def __init__(self, ctx: PostFXContext):
        self.ctx, self.lib = ctx, ctx.lib
        self.handle = ctypes.c_void_p()
        B.check(getattr(self.lib, f"mifx_{self._prefix}_create")(ctx.handle, ctypes.byref(self.handle)))

    def close(self):
        if self.handle:
            getattr(self.lib, f"mifx_{self._prefix}_destroy")(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prepare_resources(self, feature_flags=0):
        B.check(getattr(self.lib, f"mifx_{self._prefix}_prepare")(self.handle, self.ctx.handle, ctypes.c_uint32(feature_flags)))

    def reset_history(self):
        B.check(getattr(self.lib, f"mifx_{self._prefix}_reset_history")(self.handle))

    def get_intermediate(self, name):
        d = B.Image2D()
        B.check(getattr(self.lib, f"mifx_{self._prefix}_get_intermediate")(self.handle, name.encode(), ctypes.byref(d)))
        return _view(d, self.ctx.device)

    def _output(self, *args):
        d = B.Image2D()
        B.check(getattr(self.lib, f"mifx_{self._prefix}_get_output")(self.handle, *args, ctypes.byref(d)))
        return _view(d, self.ctx.device)


class ScreenSpaceAmbientOcclusion(_Effect):
    """== Diligent::ScreenSpaceAmbientOcclusion (ScreenSpaceAmbientOcclusion.hpp:57-262)."""

    _prefix = "ssao"

    def execute(self, depth, normal, attribs: B.SSAOAttribs):
        i = [B.image(depth), B.image(normal)]
        ra = B.SSAORenderAttribs(self.ctx.handle, ctypes.pointer(i[0]), ctypes.pointer(i[1]), ctypes.pointer(attribs))
        return B.check(self.lib.mifx_ssao_execute(self.handle, ctypes.byref(ra)))

    def get_ambient_occlusion(self):
        return self._output()

    def set_fused_resolve(self, enable):
        """Test hook (mifx_debug_ssao_set_fused_resolve): A7 + A8 as one resolve over work lists (default) or as two full-frame passes."""
        B.check(self.lib.mifx_debug_ssao_set_fused_resolve(self.handle, ctypes.c_int32(1 if enable else 0)))

    def export_history(self):
        """(resolved AO, history length, frame index) the next frame would reproject (mifx_ssao_export_history)."""
        return _export_history(self, (((), "ao"), ((), "history_len")))

    def import_history(self, ao, history_length, frame_index):
        _import_history(self, (ao, history_length), frame_index)


class ScreenSpaceReflection(_Effect):
    """== Diligent::ScreenSpaceReflection (ScreenSpaceReflection.hpp:62-250)."""

    _prefix = "ssr"

    def execute(self, color, depth, normal, material, motion, attribs: B.SSRAttribs):
        i = [B.image(t) for t in (color, depth, normal, material, motion)]
        ra = B.SSRRenderAttribs(self.ctx.handle, *[ctypes.pointer(x) for x in i], ctypes.pointer(attribs))
        return B.check(self.lib.mifx_ssr_execute(self.handle, ctypes.byref(ra)))

    def get_ssr_radiance(self):
        return self._output()

    def export_history(self):
        """(accumulated radiance, variance, frame index) (mifx_ssr_export_history)."""
        return _export_history(self, (((4,), "colour"), ((), "variance")))

    def import_history(self, radiance, variance, frame_index):
        _import_history(self, (radiance, variance), frame_index)


class Bloom(_Effect):
    """== Diligent::Bloom (Bloom.hpp:58-150)."""

    _prefix = "bloom"

    def execute(self, color, attribs: B.BloomAttribs):
        i = B.image(color)
        ra = B.BloomRenderAttribs(self.ctx.handle, ctypes.pointer(i), ctypes.pointer(attribs))
        return B.check(self.lib.mifx_bloom_execute(self.handle, ctypes.byref(ra)))

    def get_bloom_texture(self):
        return self._output()


class DepthOfField(_Effect):
    """== Diligent::DepthOfField (PostProcess/DepthOfField/interface/DepthOfField.hpp:52-240)."""

    _prefix = "dof"
    FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING, FEATURE_FLAG_ENABLE_KARIS_INVERSE = 1, 2

    def reset_history(self):
        """mifx_dof_reset_history (no reference counterpart): the temporal circle of confusion cleared as when the targets are created."""
        B.check(self.lib.mifx_dof_reset_history(self.handle))

    def execute(self, color, depth, attribs: B.DOFAttribs):
        i = [B.image(color), B.image(depth)]
        ra = B.DOFRenderAttribs(self.ctx.handle, ctypes.pointer(i[0]), ctypes.pointer(i[1]), ctypes.pointer(attribs))
        return B.check(self.lib.mifx_dof_execute(self.handle, ctypes.byref(ra)))

    def get_depth_of_field_texture(self):
        return self._output()

    def debug_set_last_pass(self, last_pass):
        B.check(self.lib.mifx_debug_dof_set_last_pass(self.handle, ctypes.c_uint32(last_pass)))

    @staticmethod
    def generate_kernel_points(ring_count, ring_density):
        import numpy as np

        out = (ctypes.c_float * 256)()
        n = ctypes.c_uint32(0)
        B.check(B.load().mifx_dof_generate_kernel_points(ctypes.c_int32(ring_count), ctypes.c_int32(ring_density), out, ctypes.c_uint32(128), ctypes.byref(n)))
        return np.array(out[:2 * n.value], np.float32).reshape(-1, 2)


class TemporalAntiAliasing(_Effect):
    """== Diligent::TemporalAntiAliasing (TemporalAntiAliasing.hpp:60-214)."""

    _prefix = "taa"
    FEATURE_FLAG_GAUSSIAN_WEIGHTING, FEATURE_FLAG_BICUBIC_FILTER, FEATURE_FLAG_YCOCG_COLOR_SPACE = 1, 2, 4

    def execute(self, color, attribs: B.TAAAttribs):
        i = B.image(color)
        ra = B.TAARenderAttribs(self.ctx.handle, ctypes.pointer(i), ctypes.pointer(attribs))
        return B.check(self.lib.mifx_taa_execute(self.handle, ctypes.byref(ra)))

    def get_accumulated_frame(self, is_prev_frame=False):
        return self._output(ctypes.c_int32(1 if is_prev_frame else 0))

    def export_history(self):
        """(accumulation buffer, frame index) (mifx_taa_export_history)."""
        return _export_history(self, (((4,), "colour"),))

    def import_history(self, color, frame_index):
        _import_history(self, (color,), frame_index)

    @staticmethod
    def get_jitter_offset(frame_index, width, height):
        out = (ctypes.c_float * 2)()
        B.check(B.load().mifx_taa_get_jitter_offset(ctypes.c_uint32(frame_index), ctypes.c_uint32(width), ctypes.c_uint32(height), out))
        return out[0], out[1]


class AutoExposure:
    """Average scene luminance for ToneMap (mifx_autoexposure_*): the low-resolution luminance / mip chain / UpdateAverageLuminance sequence
    of the reference's light-scattering post-process (EpipolarLightScattering.cpp:2496-2506)."""

    def __init__(self, ctx: PostFXContext):
        self.ctx, self.lib = ctx, ctx.lib
        self.handle = ctypes.c_void_p()
        B.check(self.lib.mifx_autoexposure_create(ctx.handle, ctypes.byref(self.handle)))

    def close(self):
        if self.handle:
            self.lib.mifx_autoexposure_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def execute(self, scene_color, elapsed_time_s, light_adaptation=True):
        i = B.image(scene_color)
        B.check(self.lib.mifx_autoexposure_execute(self.handle, ctypes.byref(i), ctypes.c_float(elapsed_time_s), ctypes.c_int32(1 if light_adaptation else 0)))

    def reset(self, value=0.1):
        B.check(self.lib.mifx_autoexposure_reset(self.handle, ctypes.c_float(value)))

    def plane(self, name):
        d = B.Image2D()
        B.check(self.lib.mifx_autoexposure_get_plane(self.handle, name.encode(), ctypes.byref(d)))
        return _view(d, self.ctx.device)

    def average(self):
        """GetAverageSceneLuminance(): max(0.05, average); waits for the stream."""
        v = ctypes.c_float(0.0)
        B.check(self.lib.mifx_autoexposure_get_average(self.handle, ctypes.byref(v)))
        return v.value

    def tone_map(self, hdr, attribs: B.ToneMappingAttribs, flags=0, out=None):
        """ToneMap() with fAveLogLum read from the auto-exposure plane on the device."""
        if out is None:
            out = torch.empty_like(hdr)
        i, o = B.image(hdr), B.image(out)
        B.check(self.lib.mifx_tonemap_execute_auto(self.ctx.handle, ctypes.byref(i), ctypes.byref(o), ctypes.byref(attribs), self.handle, ctypes.c_uint32(flags)))
        return out


class Comm:
    """One rank's endpoint of the sharded chain (mifx_comm): RCCL over xGMI, or an in-process group for tests on one GPU."""

    COMM_ID_BYTES = 128

    def __init__(self, lib, handle):
        self.lib, self.handle = lib, handle

    @staticmethod
    def unique_id() -> bytes:
        buf = (ctypes.c_uint8 * Comm.COMM_ID_BYTES)()
        B.check(B.load().mifx_comm_get_unique_id(buf))
        return bytes(buf)

    @classmethod
    def create(cls, ctx: "PostFXContext", uid: bytes, rank, world):
        h = ctypes.c_void_p()
        buf = (ctypes.c_uint8 * cls.COMM_ID_BYTES).from_buffer_copy(uid)
        B.check(ctx.lib.mifx_comm_create(ctx.handle, buf, ctypes.c_int32(rank), ctypes.c_int32(world), ctypes.byref(h)))
        return cls(ctx.lib, h)

    @classmethod
    def local_group(cls, ctx: "PostFXContext", world):
        arr = (ctypes.c_void_p * world)()
        B.check(ctx.lib.mifx_comm_create_local_group(ctx.handle, ctypes.c_int32(world), arr))
        return [cls(ctx.lib, ctypes.c_void_p(arr[i])) for i in range(world)]

    def info(self):
        r, w, k = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
        B.check(self.lib.mifx_comm_get_info(self.handle, ctypes.byref(r), ctypes.byref(w), ctypes.byref(k)))
        return r.value, w.value, bool(k.value)

    def set_timing(self, enable):
        """mifx_comm_set_timing: bracket every exchange group of the frames that follow with timing events (read by stats())."""
        B.check(self.lib.mifx_comm_set_timing(self.handle, ctypes.c_int32(1 if enable else 0)))

    def stats(self):
        """mifx_comm_get_stats as a dict: what the endpoint is (ranks_in_communicator = ncclCommCount of the RCCL communicator), what it has moved since creation, and the
        exchange durations recorded since set_timing(True) (waits for those groups)."""
        st = B.CommStats()
        B.check(self.lib.mifx_comm_get_stats(self.handle, ctypes.byref(st)))
        return {k: getattr(st, k) for k, _ in B.CommStats._fields_}

    def self_test(self, ctx: "PostFXContext", bytes_per_peer=1 << 16, timeout_ms=30000):
        """mifx_comm_self_test (collective): a known slab to and from every peer through the transport the frames use; raises MifxError with the transport's message."""
        B.check(self.lib.mifx_comm_self_test(self.handle, ctx.handle, ctypes.c_uint32(bytes_per_peer), ctypes.c_uint32(timeout_ms)))

    def close(self):
        if self.handle:
            self.lib.mifx_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


class Chain:
    """The canonical caller of the hot path (== HnPostProcessTask, Hydrogent/src/Tasks/HnPostProcessTask.cpp:743-948):
    PBR shade -> PostFX prep -> SSR -> SSAO -> composite -> TAA -> Bloom -> ToneMap, one mifx_chain_execute per frame."""

    def __init__(self, device=0, sobol_256d=None, scrambling_tile=None):
        self.lib = B.load()
        self.device = torch.device("cuda", device) if not isinstance(device, torch.device) else device
        dev = B.DeviceDesc(self.device.index or 0, _stream_ptr(self.device))
        info = B.PostFXCreateInfo()
        self._keep = []
        if sobol_256d is not None:
            s, t = bytes(bytearray(sobol_256d)), bytes(bytearray(scrambling_tile))
            sb, tb = ctypes.create_string_buffer(s, len(s)), ctypes.create_string_buffer(t, len(t))
            self._keep += [sb, tb]
            info.sobol_256d, info.scrambling_tile = ctypes.cast(sb, ctypes.c_void_p), ctypes.cast(tb, ctypes.c_void_p)
        self.handle = ctypes.c_void_p()
        B.check(self.lib.mifx_chain_create(ctypes.byref(dev), ctypes.byref(info), ctypes.byref(self.handle)))
        pf = ctypes.c_void_p()
        B.check(self.lib.mifx_chain_get_postfx(self.handle, ctypes.byref(pf)))
        # a non-owning PostFXContext view (IBL precompute, stream updates)
        self.postfx = PostFXContext.__new__(PostFXContext)
        self.postfx.lib, self.postfx.device, self.postfx.handle, self.postfx._keep, self.postfx.frame = self.lib, self.device, pf, [], None
        self.postfx.close = lambda: None
        # per-frame attribs with the reference defaults; callers may edit them
        self.ssao_attribs, self.ssr_attribs = B.SSAOAttribs.default(), B.SSRAttribs.default()
        self.taa_attribs, self.bloom_attribs = B.TAAAttribs.default(), B.BloomAttribs.default()
        self.tone_mapping = B.ToneMappingAttribs.default(4)
        self.taa_flags = TemporalAntiAliasing.FEATURE_FLAG_BICUBIC_FILTER  # Hydrogent default (HnPostProcessTask.hpp:109)
        self.ave_log_lum, self.ssr_scale, self.ssao_scale = 0.3, 1.0, 1.0
        self.tonemap_flags = 1  # CONVERT_OUTPUT_TO_SRGB
        self.background = (0.02, 0.03, 0.05, 0.0)

    def close(self):
        if self.handle:
            self.lib.mifx_chain_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset_history(self):
        B.check(self.lib.mifx_chain_reset_history(self.handle))

    def bind_frame(self, frame_index, g: dict, ibl: "IBLResources", shade_attribs: B.PBRShadeAttribs, out):
        """Builds the mifx_chain_frame for a G-buffer dict (synth.make_frame layout). Returns an opaque object for execute()."""
        h, w = g["depth"].shape
        g = dict(g)
        for k in ("base_color", "normal", "material", "emissive"):  # 4-channel planes in the storage of this build (float16 with MIFX_STORAGE=h4; a no-op otherwise)
            if g.get(k) is not None and g[k].dtype != B.storage_dtype():
                g[k] = B.to_storage(g[k])
        imgs = {k: B.image(g[k]) for k in ("base_color", "normal", "material", "depth", "motion", "prev_depth")}
        for k in ("emissive", "occlusion"):
            if g.get(k) is not None:
                imgs[k] = B.image(g[k])
        p = lambda k: ctypes.pointer(imgs[k]) if k in imgs else None  # noqa: E731
        f = B.ChainFrame()
        f.frame = B.FrameDesc(frame_index, w, h, w, h)
        f.gbuffer = B.GBuffer(p("base_color"), p("normal"), p("material"), p("depth"), p("emissive"), p("occlusion"))
        f.motion, f.prev_depth = p("motion"), p("prev_depth")
        f.curr_camera, f.prev_camera = ctypes.pointer(g["camera"]), ctypes.pointer(g["prev_camera"])
        f.ibl, f.pbr = ctypes.pointer(ibl.struct), ctypes.pointer(shade_attribs)
        f.ssao, f.ssr = ctypes.pointer(self.ssao_attribs), ctypes.pointer(self.ssr_attribs)
        f.taa, f.bloom = ctypes.pointer(self.taa_attribs), ctypes.pointer(self.bloom_attribs)
        f.tone_mapping = ctypes.pointer(self.tone_mapping)
        f.ave_log_lum, f.ssr_scale, f.ssao_scale = self.ave_log_lum, self.ssr_scale, self.ssao_scale
        f.background[:] = list(self.background)
        f.taa_feature_flags, f.tonemap_flags = self.taa_flags, self.tonemap_flags
        o = B.image(out)
        return (f, o, imgs, g, ibl, shade_attribs, out)

    def execute(self, bound):
        B.check(self.lib.mifx_postfx_set_stream(self.postfx.handle, _stream_ptr(self.device)))
        self._last_bound = bound  # (effect_output("ssr") may have to run the deferred cleanup on this frame's depth / normal planes)
        return B.check(self.lib.mifx_chain_execute(self.handle, ctypes.byref(bound[0]), ctypes.byref(bound[1])))

    def execute_native(self, bound, fmt: str, pitch_bytes=None):
        """mifx_chain_execute_native: the frame in the copy-frame target's own format, torch.uint8 (H, pitch_bytes)."""
        out = bound[-1]
        raw, dst = _native_target(self.postfx, out.shape[0], out.shape[1], fmt, pitch_bytes, out.device)
        B.check(self.lib.mifx_postfx_set_stream(self.postfx.handle, _stream_ptr(self.device)))
        self._last_bound = bound
        B.check(self.lib.mifx_chain_execute_native(self.handle, ctypes.byref(bound[0]), ctypes.byref(dst)))
        return raw

    def set_auto_exposure(self, enable, elapsed_time_s=1.0 / 60.0, light_adaptation=True):
        """The final tone map takes fAveLogLum from the adapted average luminance of the Bloom output instead of self.ave_log_lum."""
        B.check(self.lib.mifx_chain_set_auto_exposure(self.handle, ctypes.c_int32(1 if enable else 0), ctypes.c_float(elapsed_time_s), ctypes.c_int32(1 if light_adaptation else 0)))
        self.auto_exposure = bool(enable)  # (sharded.py: the luminance rows are exchanged after phase 3)

    def effect_output(self, name):
        """Output plane of one of the chain's own effect objects: "ssao", "ssr", "taa", "bloom", "dof" (a view, valid until the next prepare)."""
        h = ctypes.c_void_p()
        B.check(self.lib.mifx_chain_get_effect(self.handle, name.encode(), ctypes.byref(h)))
        if not h:
            raise ValueError(f"the chain has no '{name}' effect (not enabled)")
        d = B.Image2D()
        extra = (ctypes.c_int32(0),) if name == "taa" else ()
        if name == "ssr":
            if getattr(self, "_last_bound", None) is None:
                raise RuntimeError("Chain.effect_output('ssr'): no frame has been executed through this object yet (the deferred cleanup needs its depth / normal planes)")
            # the chain's composite evaluated the bilateral cleanup itself (MIFX_CHAIN_FUSE_SSR_CLEANUP_INTO_COMPOSITE): produce the plane from the frame just executed
            imgs = self._last_bound[2]
            B.check(self.lib.mifx_ssr_run_deferred_cleanup(h, ctypes.byref(imgs["depth"]), ctypes.byref(imgs["normal"])))
        B.check(getattr(self.lib, f"mifx_{name}_get_output")(h, *extra, ctypes.byref(d)))
        return _view(d, self.device)

    FUSE_TONE_MAP_INTO_BLOOM, FUSE_SSR_MASK_INTO_SHADE, FUSE_SSR_CLEANUP_INTO_COMPOSITE, FUSE_SSAO_RESOLVE, FUSE_BLOOM_OUTPUT_ON_DEMAND, FUSE_COMPOSITE_INTO_TAA, FUSE_DEFAULT, FUSE_ALL, FUSE_EXPERIMENTAL, FUSE_EVERY_SWITCH = 1, 2, 4, 8, 16, 32, 31, 31, 32, 63

    def set_fusion_mask(self, mask):
        """mifx_chain_set_fusion_mask: every fusion switch of the chain (MIFX_CHAIN_FUSE_*; FUSE_DEFAULT = what a new chain has; the results are bit-identical either way)."""
        B.check(self.lib.mifx_chain_set_fusion_mask(self.handle, ctypes.c_uint32(mask)))

    def set_overlap(self, mode):
        """mifx_chain_set_overlap: 0 = one stream; 1 = PostFX prep + SSAO on a second stream beside the shade + SSR; 3 = three lanes (shade + prep + Hi-Z + SSAO | SSR +
        composite + TAA | Bloom) sliding across frames; 2 = also across frames (the next frame's prep +
        SSAO start as soon as this frame's TAA is done, under the Bloom pyramid) -- mode 2 requires that a frame's input planes are complete when execute is called;
        4 = the lanes of 3 with two frames in flight (the next frame's shade + SSAO beside this frame's SSR resolve / composite / TAA); 5 = 4 with the composite, TAA and depth
        of field on the Bloom lane (the next frame's ray march beside them)."""
        B.check(self.lib.mifx_chain_set_overlap(self.handle, ctypes.c_int32(int(mode))))

    def set_lane_edges(self, edges):
        """mifx_chain_set_lane_edges: "waiter<signal@frames,..." (mode 4: extra ordering between kernels of different lanes / frames; results do not depend on it)."""
        B.check(self.lib.mifx_chain_set_lane_edges(self.handle, (edges or "").encode()))

    def set_fusion(self, tone_map_into_bloom=True, ssr_mask_into_shade=True):
        """mifx_chain_set_fusion: pass fusion inside the chain (bit-identical results; on by default)."""
        B.check(self.lib.mifx_chain_set_fusion(self.handle, ctypes.c_int32(1 if tone_map_into_bloom else 0), ctypes.c_int32(1 if ssr_mask_into_shade else 0)))

    def effect(self, name):
        """Non-owning view of one of the chain's own effect objects ("ssao", "ssr", "taa", "bloom"): intermediates, history export / import."""
        cls = {"ssao": ScreenSpaceAmbientOcclusion, "ssr": ScreenSpaceReflection, "taa": TemporalAntiAliasing, "bloom": Bloom}[name]
        h = ctypes.c_void_p()
        B.check(self.lib.mifx_chain_get_effect(self.handle, name.encode(), ctypes.byref(h)))
        fx = cls.__new__(cls)
        fx.ctx, fx.lib, fx.handle = self.postfx, self.lib, h
        fx.close = lambda: None  # the chain owns it
        return fx

    def set_effect_feature_flags(self, ssao_feature_flags=0, ssr_feature_flags=0):
        """FEATURE_FLAGS of the chain's SSAO / SSR objects (SSAO 2 = HALF_RESOLUTION, SSR 1 = PREVIOUS_FRAME)."""
        B.check(self.lib.mifx_chain_set_effect_feature_flags(self.handle, ctypes.c_uint32(ssao_feature_flags), ctypes.c_uint32(ssr_feature_flags)))

    def set_postfx_feature_flags(self, feature_flags):
        """PostFXContext::FEATURE_FLAGS of the chain's context (1 = FEATURE_FLAG_REVERSED_DEPTH)."""
        B.check(self.lib.mifx_chain_set_postfx_feature_flags(self.handle, ctypes.c_uint32(feature_flags)))

    def set_depth_of_field(self, attribs: "B.DOFAttribs | None", feature_flags=0):
        """DepthOfField::Execute between TAA and Bloom (HnPostProcessTask.cpp:899-909); None turns it off."""
        B.check(self.lib.mifx_chain_set_depth_of_field(self.handle, ctypes.byref(attribs) if attribs is not None else None, ctypes.c_uint32(feature_flags)))

    def set_material_layers(self, layers: "dict | None" = None, flags=0, iridescence_ior=1.3, anisotropy_rotation=0.0, shadows=None):
        """The chain's shade with material layers / shadow-mapped lights (mifx_chain_set_material_layers); arguments as pbr_shade_layers().  The tensors are kept alive here."""
        ly = sh = None
        self._layer_keep = (layers, shadows)
        if layers is not None and flags:
            imgs = {k: B.image(layers[k]) for k in B.PBR_LAYER_PLANES if layers.get(k) is not None}
            ly = B.PBRLayers(flags, iridescence_ior, anisotropy_rotation, 0, *[ctypes.pointer(imgs[k]) if k in imgs else None for k in B.PBR_LAYER_PLANES])
        if shadows is not None:
            sm, infos, pcf = shadows
            assert sm.dtype == torch.float32 and sm.dim() == 3 and sm.is_contiguous()
            arr = B.ShadowMapArray(sm.data_ptr(), sm.shape[2], sm.shape[1], sm.shape[0], sm.stride(1) * 4, sm.stride(0) * 4)
            rows = (B.PBRShadowMapInfo * len(infos))()
            for i, r in enumerate(infos):
                rows[i] = r if isinstance(r, B.PBRShadowMapInfo) else B.PBRShadowMapInfo.from_buffer_copy(r.astype("float32").tobytes())
            sh = B.PBRShadows(ctypes.pointer(arr), ctypes.cast(rows, ctypes.POINTER(B.PBRShadowMapInfo)), len(infos), pcf)
        B.check(self.lib.mifx_chain_set_material_layers(self.handle, ctypes.byref(ly) if ly is not None else None, ctypes.byref(sh) if sh is not None else None))

    def auto_exposure_average(self):
        h = ctypes.c_void_p()
        B.check(self.lib.mifx_chain_get_auto_exposure(self.handle, ctypes.byref(h)))
        if not h:
            return None
        v = ctypes.c_float(0.0)
        B.check(self.lib.mifx_autoexposure_get_average(h, ctypes.byref(v)))
        return v.value

    # ---- row-band sharding with the exchanges inside the library (mifx_comm_* / mifx_chain_set_sharding / mifx_chain_execute_sharded)
    def set_sharding(self, comm: "Comm | None", row_cuts=None, max_motion_rows=0):
        if comm is None:
            B.check(self.lib.mifx_chain_set_sharding(self.handle, None, None, ctypes.c_int32(0)))
            return
        cuts = (ctypes.c_int32 * len(row_cuts))(*row_cuts)
        B.check(self.lib.mifx_chain_set_sharding(self.handle, comm.handle, cuts, ctypes.c_int32(max_motion_rows)))
        self._keep.append(comm)

    def execute_sharded(self, bound):
        B.check(self.lib.mifx_postfx_set_stream(self.postfx.handle, _stream_ptr(self.device)))
        self._last_bound = bound
        return B.check(self.lib.mifx_chain_execute_sharded(self.handle, ctypes.byref(bound[0]), ctypes.byref(bound[1])))

    def execute_band(self, bound):
        """mifx_chain_execute_band: the phases (and lanes) of execute_sharded for the band of set_row_band, exchanges left out -- one rank's compute side (tools)."""
        B.check(self.lib.mifx_postfx_set_stream(self.postfx.handle, _stream_ptr(self.device)))
        self._last_bound = bound
        return B.check(self.lib.mifx_chain_execute_band(self.handle, ctypes.byref(bound[0]), ctypes.byref(bound[1])))

    # ---- row-band sharding (mifx_chain_set_row_band / execute_phase / get_shard_info / get_shard_plane)
    def set_row_band(self, row_begin, row_end, max_motion_rows):
        B.check(self.lib.mifx_chain_set_row_band(self.handle, ctypes.c_int32(row_begin), ctypes.c_int32(row_end), ctypes.c_int32(max_motion_rows)))

    def execute_phase(self, bound, phase):
        B.check(self.lib.mifx_postfx_set_stream(self.postfx.handle, _stream_ptr(self.device)))
        self._last_bound = bound
        return B.check(self.lib.mifx_chain_execute_phase(self.handle, ctypes.byref(bound[0]), ctypes.byref(bound[1]), ctypes.c_int32(phase)))

    def shard_info(self, bound):
        info = B.ShardInfo()
        B.check(self.lib.mifx_chain_get_shard_info(self.handle, ctypes.byref(bound[0]), ctypes.byref(info)))
        return info

    def shard_plane_image(self, name):
        """The same plane as an image view (H, W[, C]) in its own texel type."""
        d = B.Image2D()
        B.check(self.lib.mifx_chain_get_shard_plane(self.handle, name.encode(), ctypes.byref(d)))
        return _view(d, self.device)

    def shard_plane(self, name):
        """Contiguous (height, pitch_floats) torch view (no copy) of one of the planes that move between ranks: a block of rows is one
        contiguous slab, padding included.  Valid until the next prepare that changes the size."""
        d = B.Image2D()
        B.check(self.lib.mifx_chain_get_shard_plane(self.handle, name.encode(), ctypes.byref(d)))
        pitch_f = d.pitch_bytes // 4

        class _Holder:
            pass

        h = _Holder()
        h.__cuda_array_interface__ = {"shape": (d.height * pitch_f,), "typestr": "<f4", "data": (d.data, False), "version": 2}
        return torch.as_tensor(h, device=self.device).view(d.height, pitch_f)
