// mifx_core.cpp -- status strings, thread-local error detail, image validation, owned planes, camera packing.
#include <dlfcn.h>
#include <cstdlib>
#include "mifx_host.h"

namespace mifx
{
static thread_local char g_last_error[512] = "";

void set_error(const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

unsigned occupancy_pad_from_env(const char* name)
{
    const char* e = std::getenv(name);
    if (!e || !e[0]) return 0u;
    const long v = std::atol(e);
    return v <= 0 ? 0u : v > 60 * 1024 ? 60u * 1024u : unsigned(v);
}

// ------------------------------------------------------------------------------------------------ rocTX ranges (MifxRange, mifx_host.h)
namespace
{
struct Roctx
{
    int  state = 0; // 0: not tried, 1: loaded, -1: unavailable
    int (*push)(const char*) = nullptr;
    int (*pop)()             = nullptr;
};
Roctx g_roctx;
int   g_markers = -1; // -1: follow the environment (MIFX_ROCTX), 0 / 1: mifx_set_markers
bool markers_enabled()
{
    if (g_markers < 0)
    {
        const char* e = std::getenv("MIFX_ROCTX");
        g_markers     = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    if (g_markers == 0) return false;
    if (g_roctx.state == 0)
    {
        g_roctx.state = -1;
        for (const char* name : {"libroctx64.so.4", "libroctx64.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/libroctx64.so"})
            if (void* lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL))
            {
                g_roctx.push = reinterpret_cast<int (*)(const char*)>(dlsym(lib, "roctxRangePushA"));
                g_roctx.pop  = reinterpret_cast<int (*)()>(dlsym(lib, "roctxRangePop"));
                if (g_roctx.push && g_roctx.pop) { g_roctx.state = 1; break; }
            }
    }
    return g_roctx.state == 1;
}
} // namespace
MifxRange::MifxRange(const char* name) : on(markers_enabled())
{
    if (on) (void)g_roctx.push(name);
}
void MifxRange::end()
{
    if (on) (void)g_roctx.pop();
    on = false;
}
void set_markers(int enable) { g_markers = enable ? 1 : 0; }
} // namespace mifx

// kernel -> the marker of the reference pass it implements (ScopedDebugGroup names of PostFXContext.cpp, ScreenSpaceAmbientOcclusion.cpp, ScreenSpaceReflection.cpp,
// TemporalAntiAliasing.cpp, Bloom.cpp, DepthOfField.cpp; the shade, composite and copy-frame draws carry no marker in the reference and keep descriptive names)
const char* mifx_reference_pass_name(const char* kernel)
{
    static const struct { const char* kernel; const char* pass; } table[] = {
        {"postfx_prep_kernel", "ComputeReprojectedDepth + ComputeClosestMotion"},
        {"ssao_compute_ao_kernel", "ComputeAmbientOcclusion"}, {"ssao_temporal_kernel", "ComputeTemporalAccumulation"}, {"ssao_resample_kernel", "ComputeResampledHistory"},
        {"ssao_spatial_kernel", "ComputeSpatialReconstruction"}, {"ssao_resolve_list_kernels", "ComputeResampledHistory + ComputeSpatialReconstruction"}, {"ssr_mask_roughness_kernel", "ComputeStencilMaskAndExtractRoughness"},
        {"ssr_intersection_kernel", "ComputeIntersection"}, {"ssr_spatial_kernel", "ComputeSpatialReconstruction"}, {"ssr_temporal_kernel", "ComputeTemporalAccumulation"},
        {"ssr_bilateral_kernel", "ComputeBilateralCleanup"}, {"bloom_prefilter_kernel", "ComputePrefilteredTexture"}, {"bloom_upsample_kernel", "ComputeUpsampledTexture"},
        {"bloom_upsample_tonemap_kernel", "ComputeUpsampledTexture + CopyFrame ToneMap"}, {"taa_kernel", "ComputeTemporalAccumulation"},
        {"dof_bokeh_gather_kernel", "ComputeBokehFirstPass"}, {"pbr_shade_kernel", "RenderPBR shade"},
        {"pbr_shade_ssr_mask_kernel", "RenderPBR shade + ComputeStencilMaskAndExtractRoughness"}, {"composite_kernel", "HnPostProcess composite"}, {"tonemap_kernel", "CopyFrame ToneMap"}};
    for (const auto& e : table)
        if (std::strcmp(e.kernel, kernel) == 0) return e.pass;
    return kernel;
}
namespace mifx
{

mifx_status to_img(const mifx_image2d* im, uint32_t fmt, const char* what, Img& out)
{
    MIFX_REQUIRE(im != nullptr, "%s: image descriptor must not be null", what);
    MIFX_REQUIRE(im->data != nullptr, "%s: data pointer must not be null", what);
    MIFX_REQUIRE(im->width > 0 && im->height > 0, "%s: empty image (%ux%u)", what, im->width, im->height);
    fmt = storage_format(fmt);
    MIFX_REQUIRE(im->format == fmt, "%s: format %u, expected %u%s", what, im->format, fmt,
                 fmt == MIFX_FORMAT_F16X4 ? " (this is the RGBA16_FLOAT storage build of the library: 4-channel images are MIFX_FORMAT_F16X4)" : "");
    const uint32_t ts = texel_size(fmt);
    MIFX_REQUIRE(im->pitch_bytes >= im->width * ts && im->pitch_bytes % ts == 0, "%s: bad pitch %u for width %u", what, im->pitch_bytes, im->width);
    MIFX_REQUIRE((reinterpret_cast<uintptr_t>(im->data) % ts) == 0, "%s: data pointer not aligned to the texel size %u", what, ts);
    // the kernels address texels with 32-bit byte offsets from the plane pointer (24-bit row x pitch products, mifx_device.h: bilinear_taps)
    MIFX_REQUIRE(im->pitch_bytes < (1u << 24) && im->height < (1u << 24) && uint64_t(im->pitch_bytes) * im->height <= 0xFFFFFFFFull,
                 "%s: plane of %u rows x %u bytes exceeds the 4 GiB addressing limit of one plane", what, im->height, im->pitch_bytes);
    out = Img{static_cast<unsigned char*>(im->data), int(im->width), int(im->height), int(im->pitch_bytes), 0, 0}; // all rows
    return MIFX_OK;
}

mifx_status to_img_hdr(const mifx_image2d* im, const char* what, Img& out, bool& packed)
{
    packed = im != nullptr && im->format == MIFX_FORMAT_R11G11B10 && storage_format(MIFX_PLANE_BLOOM) == MIFX_FORMAT_R11G11B10;
    return to_img(im, packed ? uint32_t(MIFX_PLANE_BLOOM) : uint32_t(MIFX_FORMAT_F32X4), what, out);
}

mifx_status to_img_wh(const mifx_image2d* im, uint32_t fmt, uint32_t w, uint32_t h, const char* what, Img& out)
{
    MIFX_CHECK(to_img(im, fmt, what, out));
    MIFX_REQUIRE(im->width == w && im->height == h, "%s: size %ux%u, expected %ux%u", what, im->width, im->height, w, h);
    return MIFX_OK;
}

CamK make_camk(const mifx_camera_attribs& c, bool reversedDepth)
{
    CamK k;
    std::memcpy(k.view.m, c.mView, 64);
    std::memcpy(k.proj.m, c.mProj, 64);
    std::memcpy(k.viewProj.m, c.mViewProj, 64);
    std::memcpy(k.viewInv.m, c.mViewInv, 64);
    std::memcpy(k.viewProjInv.m, c.mViewProjInv, 64);
    k.pos[0] = c.f4Position[0]; k.pos[1] = c.f4Position[1]; k.pos[2] = c.f4Position[2];
    k.vw = c.f4ViewportSize[0]; k.vh = c.f4ViewportSize[1]; k.ivw = c.f4ViewportSize[2]; k.ivh = c.f4ViewportSize[3];
    k.jx = c.f2Jitter[0]; k.jy = c.f2Jitter[1];
    k.frameIndex = c.uiFrameIndex;
    k.reversedDepth = reversedDepth ? 1 : 0;
    return k;
}

mifx_status Plane::alloc(uint32_t width, uint32_t height, uint32_t format)
{
    format = storage_format(format);
    if (data && w == width && h == height && fmt == format) return MIFX_OK;
    release();
    const uint32_t ts = texel_size(format);
    MIFX_REQUIRE(ts != 0 && width > 0 && height > 0, "Plane::alloc: bad arguments %ux%u fmt %u", width, height, format);
    MIFX_REQUIRE(uint64_t(width) * ts + 255u < (1u << 24) && height < (1u << 24), "Plane::alloc: %ux%u is beyond the supported plane size", width, height);
    const uint32_t p = ((width * ts + 255u) / 256u) * 256u;
    const size_t   n = size_t(p) * height;
    MIFX_REQUIRE(n <= 0xFFFFFFFFull, "Plane::alloc: %ux%u fmt %u exceeds the 4 GiB addressing limit of one plane", width, height, format);
    void* ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, n);
    if (e != hipSuccess)
    {
        set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
        return MIFX_ERR_OUT_OF_MEMORY;
    }
    data = ptr; w = width; h = height; pitch = p; fmt = format; bytes = n;
    return MIFX_OK;
}

void Plane::release()
{
    if (data && owned) (void)hipFree(data);
    data = nullptr; w = h = pitch = fmt = 0; bytes = 0; owned = true;
}

mifx_status Plane::fill(hipStream_t s, float value) const
{
    if (!data) return MIFX_OK;
    if (value == 0.0f)
    {
        MIFX_HIP_CHECK(hipMemsetAsync(data, 0, bytes, s));
        return MIFX_OK;
    }
    if (fmt == MIFX_FORMAT_U8 || fmt == MIFX_FORMAT_F16)
    {
        // the history targets of the native-storage build are cleared to 1.0: code 255 / binary16 0x3C00 (pitched rows: every byte / half of the allocation may take the value)
        MIFX_REQUIRE(value == 1.0f, "Plane::fill: a narrow plane is only ever cleared to 0 or 1");
        if (fmt == MIFX_FORMAT_U8) MIFX_HIP_CHECK(hipMemsetAsync(data, 0xFF, bytes, s));
        else MIFX_HIP_CHECK(hipMemsetD16Async(data, 0x3C00, bytes / 2u, s));
        return MIFX_OK;
    }
    if (fmt != MIFX_FORMAT_F32 && fmt != MIFX_FORMAT_F32X2 && fmt != MIFX_FORMAT_F32X4)
    {
        set_error("Plane::fill: a plane of format %u is only ever cleared to 0", fmt);
        return MIFX_ERR_INVALID_ARG;
    }
    return launch_fill_f32(s, view(), int(texel_size(fmt) / 4u), value);
}
} // namespace mifx

extern "C" {
const char* mifx_status_string(mifx_status s)
{
    switch (s)
    {
        case MIFX_OK: return "MIFX_OK";
        case MIFX_NO_HISTORY: return "MIFX_NO_HISTORY";
        case MIFX_ERR_INVALID_ARG: return "MIFX_ERR_INVALID_ARG";
        case MIFX_ERR_INVALID_OP: return "MIFX_ERR_INVALID_OP";
        case MIFX_ERR_HIP: return "MIFX_ERR_HIP";
        case MIFX_ERR_OUT_OF_MEMORY: return "MIFX_ERR_OUT_OF_MEMORY";
        case MIFX_ERR_NOT_IMPLEMENTED: return "MIFX_ERR_NOT_IMPLEMENTED";
        case MIFX_ERR_COMM: return "MIFX_ERR_COMM";
        default: return "MIFX_<unknown>";
    }
}
const char* mifx_last_error(void) { return mifx::g_last_error; }
uint32_t    mifx_abi_version(void) { return 3; } // 2: mifx_pbr_shade_attribs::Workflow, history export / import, mifx_comm_*, mifx_chain_set_fusion, markers; 3: round 4's entries
                                                 // (PostFXContext helpers, MIFX_FORMAT_U16, mifx_pbr_shade_execute_layers, mifx_chain_set_material_layers, ...): additions only
void        mifx_set_markers(int32_t enable) { mifx::set_markers(enable); }
uint32_t    mifx_storage_mode(void)
{
#ifdef MIFX_STORAGE_H4
    return MIFX_STORAGE_RGBA16F;
#else
    return MIFX_STORAGE_FP32;
#endif
}
uint32_t    mifx_sizeof(const char* n)
{
    if (!n) return 0;
#define MIFX_SZ(name, type) if (std::strcmp(n, name) == 0) return uint32_t(sizeof(type))
    MIFX_SZ("image2d", mifx_image2d);
    MIFX_SZ("cubemap", mifx_cubemap);
    MIFX_SZ("camera_attribs", mifx_camera_attribs);
    MIFX_SZ("tone_mapping_attribs", mifx_tone_mapping_attribs);
    MIFX_SZ("ssao_attribs", mifx_ssao_attribs);
    MIFX_SZ("ssr_attribs", mifx_ssr_attribs);
    MIFX_SZ("bloom_attribs", mifx_bloom_attribs);
    MIFX_SZ("dof_attribs", mifx_dof_attribs);
    MIFX_SZ("taa_attribs", mifx_taa_attribs);
    MIFX_SZ("pbr_light_attribs", mifx_pbr_light_attribs);
    MIFX_SZ("pbr_shadow_map_info", mifx_pbr_shadow_map_info);
    MIFX_SZ("pbr_shade_attribs", mifx_pbr_shade_attribs);
    MIFX_SZ("pbr_renderer_shader_parameters", mifx_pbr_renderer_shader_parameters);
    MIFX_SZ("pbr_material_basic_attribs", mifx_pbr_material_basic_attribs);
    MIFX_SZ("frame_desc", mifx_frame_desc);
    MIFX_SZ("chain_frame", mifx_chain_frame);
    MIFX_SZ("composite_attribs", mifx_composite_attribs);
    MIFX_SZ("gbuffer", mifx_gbuffer);
    MIFX_SZ("ibl", mifx_ibl);
    MIFX_SZ("shard_info", mifx_shard_info);
    MIFX_SZ("comm_stats", mifx_comm_stats);
#undef MIFX_SZ
    return 0;
}
}

// layout pins against the reference structs (measured with g++ on the reference headers)
static_assert(sizeof(mifx_camera_attribs) == 576, "CameraAttribs");
static_assert(sizeof(mifx_tone_mapping_attribs) == 48, "ToneMappingAttribs");
static_assert(sizeof(mifx_ssao_attribs) == 48, "ScreenSpaceAmbientOcclusionAttribs");
static_assert(sizeof(mifx_ssr_attribs) == 48, "ScreenSpaceReflectionAttribs");
static_assert(sizeof(mifx_bloom_attribs) == 32, "BloomAttribs");
static_assert(sizeof(mifx_taa_attribs) == 16, "TemporalAntiAliasingAttribs");
static_assert(sizeof(mifx_pbr_light_attribs) == 64, "PBRLightAttribs");
static_assert(sizeof(mifx_pbr_renderer_shader_parameters) == 144, "PBRRendererShaderParameters");
static_assert(sizeof(mifx_pbr_material_basic_attribs) == 96, "PBRMaterialBasicAttribs");
