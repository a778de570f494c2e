// api_autoexposure.cpp -- C ABI of the auto-exposure pass (include/mifx.h "auto exposure") and of the tone-map entry that reads its result.
#include "mifx_objects.h"
#include <cstring>

using namespace mifx;

extern "C" {

mifx_status mifx_autoexposure_reset(mifx_autoexposure* ae, float average_luminance)
{
    MIFX_REQUIRE(ae != nullptr, "mifx_autoexposure_reset: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ae->ctx->device));
    ae->ctx->queued_outside_execute();
    return ae->average.fill(ae->ctx->stream, average_luminance);
}

mifx_status mifx_autoexposure_create(mifx_postfx* ctx, mifx_autoexposure** out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_autoexposure_create: null argument");
    *out = nullptr;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    mifx_autoexposure* ae = new mifx_autoexposure();
    ae->ctx = ctx;
    mifx_status st = ae->low_res.alloc(64, 64, MIFX_FORMAT_F32X2);
    if (st >= 0) st = ae->average.alloc(1, 1, MIFX_FORMAT_F32);
    if (st >= 0) st = mifx_autoexposure_reset(ae, 0.1f); // TexDesc.ClearValue (EpipolarLightScattering.cpp:892-905)
    if (st < 0)
    {
        delete ae;
        return st;
    }
    *out = ae;
    return MIFX_OK;
}

void mifx_autoexposure_destroy(mifx_autoexposure* ae) { delete ae; }

mifx_status mifx_autoexposure_execute(mifx_autoexposure* ae, const mifx_image2d* scene_color, float elapsed_time_s, int32_t light_adaptation)
{
    MIFX_REQUIRE(ae != nullptr && scene_color != nullptr, "mifx_autoexposure_execute: null argument");
    MIFX_REQUIRE(elapsed_time_s >= 0.0f, "mifx_autoexposure_execute: negative elapsed time");
    Img  color;
    bool packed = false;
    MIFX_CHECK(to_img_hdr(scene_color, "scene_color", color, packed));
    MIFX_HIP_CHECK(hipSetDevice(ae->ctx->device));
    return launch_autoexposure(ae->ctx->stream, color, ae->low_res.view(), static_cast<float*>(ae->average.data), elapsed_time_s, light_adaptation != 0, packed);
}

mifx_status mifx_autoexposure_get_plane(mifx_autoexposure* ae, const char* name, mifx_image2d* out)
{
    MIFX_REQUIRE(ae != nullptr && name != nullptr && out != nullptr, "mifx_autoexposure_get_plane: null argument");
    if (std::strcmp(name, "average_luminance") == 0) *out = ae->average.desc();
    else if (std::strcmp(name, "low_res_luminance") == 0) *out = ae->low_res.desc();
    else
    {
        set_error("mifx_autoexposure_get_plane: unknown plane '%s'", name);
        return MIFX_ERR_INVALID_ARG;
    }
    return MIFX_OK;
}

mifx_status mifx_autoexposure_get_average(mifx_autoexposure* ae, float* out)
{
    MIFX_REQUIRE(ae != nullptr && out != nullptr, "mifx_autoexposure_get_average: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ae->ctx->device));
    float v = 0.0f;
    MIFX_HIP_CHECK(hipMemcpyAsync(&v, ae->average.data, sizeof(float), hipMemcpyDeviceToHost, ae->ctx->stream));
    MIFX_HIP_CHECK(hipStreamSynchronize(ae->ctx->stream));
    *out = v > 0.05f ? v : 0.05f; // GetAverageSceneLuminance (AtmosphereShadersCommon.fxh:188-195)
    return MIFX_OK;
}

mifx_status mifx_tonemap_execute_auto(mifx_postfx* ctx, const mifx_image2d* hdr_in, const mifx_image2d* ldr_out, const mifx_tone_mapping_attribs* attribs, mifx_autoexposure* ae,
                                      uint32_t flags)
{
    MIFX_REQUIRE(ctx != nullptr && attribs != nullptr && ae != nullptr, "mifx_tonemap_execute_auto: null argument");
    MIFX_REQUIRE(attribs->iToneMappingMode >= 0 && attribs->iToneMappingMode <= MIFX_TONE_MAPPING_MODE_COMMERCE, "mifx_tonemap_execute_auto: unknown tone mapping mode %d",
                 attribs->iToneMappingMode);
    MIFX_REQUIRE((flags & ~uint32_t(MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB)) == 0, "mifx_tonemap_execute_auto: unknown flags 0x%x", flags);
    // the average is written on the auto-exposure object's context stream and read here on ctx's: one context, one stream, or the read races the write
    MIFX_REQUIRE(ae->ctx == ctx || (ae->ctx->device == ctx->device && ae->ctx->stream == ctx->stream),
                 "mifx_tonemap_execute_auto: the auto-exposure object belongs to a context on another device / stream");
    Img  in, out;
    bool packed = false;
    MIFX_CHECK(to_img_hdr(hdr_in, "hdr_in", in, packed));
    MIFX_CHECK(to_img_wh(ldr_out, MIFX_FORMAT_F32X4, hdr_in->width, hdr_in->height, "ldr_out", out));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_tonemap(ctx->stream, in, win(out, ctx->needed_rows(out.h)), *attribs, 1.0f, flags, static_cast<const float*>(ae->average.data), packed);
}

} // extern "C"
