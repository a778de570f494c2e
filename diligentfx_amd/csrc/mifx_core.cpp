// mifx_core.cpp -- status strings, thread-local error detail, image validation, owned planes, camera packing.
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

mifx_status to_img(const mifx_image2d* im, uint32_t fmt, const char* what, Img& out)
{
    MIFX_REQUIRE(im != nullptr, "%s: image descriptor must not be null", what);
    MIFX_REQUIRE(im->data != nullptr, "%s: data pointer must not be null", what);
    MIFX_REQUIRE(im->width > 0 && im->height > 0, "%s: empty image (%ux%u)", what, im->width, im->height);
    MIFX_REQUIRE(im->format == fmt, "%s: format %u, expected %u", what, im->format, fmt);
    const uint32_t ts = texel_size(fmt);
    MIFX_REQUIRE(im->pitch_bytes >= im->width * ts && im->pitch_bytes % ts == 0, "%s: bad pitch %u for width %u", what, im->pitch_bytes, im->width);
    MIFX_REQUIRE((reinterpret_cast<uintptr_t>(im->data) % ts) == 0, "%s: data pointer not aligned to the texel size %u", what, ts);
    // the kernels address texels with 32-bit byte offsets from the plane pointer (24-bit row x pitch products, mifx_device.h: bilinear_taps)
    MIFX_REQUIRE(im->pitch_bytes < (1u << 24) && im->height < (1u << 24) && uint64_t(im->pitch_bytes) * im->height <= 0xFFFFFFFFull,
                 "%s: plane of %u rows x %u bytes exceeds the 4 GiB addressing limit of one plane", what, im->height, im->pitch_bytes);
    out = Img{static_cast<unsigned char*>(im->data), int(im->width), int(im->height), int(im->pitch_bytes), 0, 0}; // all rows
    return MIFX_OK;
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
uint32_t    mifx_abi_version(void) { return 1; }
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
    MIFX_SZ("frame_desc", mifx_frame_desc);
    MIFX_SZ("chain_frame", mifx_chain_frame);
    MIFX_SZ("composite_attribs", mifx_composite_attribs);
    MIFX_SZ("gbuffer", mifx_gbuffer);
    MIFX_SZ("ibl", mifx_ibl);
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
