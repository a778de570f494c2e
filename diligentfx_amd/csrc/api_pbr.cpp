// api_pbr.cpp -- C ABI of the PBR shading entry and of the SSR/SSAO composite.
#include "mifx_objects.h"

using namespace mifx;

extern "C" {

mifx_status mifx_pbr_shade_execute(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_camera_attribs* camera, const mifx_pbr_shade_attribs* attribs,
                                   const mifx_ibl* ibl, const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl)
{
    MIFX_REQUIRE(ctx != nullptr && gbuffer != nullptr && camera != nullptr && attribs != nullptr && ibl != nullptr && out_radiance != nullptr,
                 "mifx_pbr_shade_execute: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_pbr_shade(ctx->stream, gbuffer, *camera, *attribs, ibl, background, out_radiance, out_specular_ibl);
}

mifx_status mifx_composite_execute(mifx_postfx* ctx, const mifx_composite_attribs* attribs, const mifx_image2d* out)
{
    MIFX_REQUIRE(ctx != nullptr && attribs != nullptr && out != nullptr, "mifx_composite_execute: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_composite(ctx->stream, *attribs, out);
}

} // extern "C"
