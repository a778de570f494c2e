// api_history.cpp -- export / import of the temporal state of the three accumulating effects (mifx_*_export_history / import_history).
//
// The reference keeps this state inside the effect objects with no way to read or write it: AO history + history length ping-pong by
// FrameDesc.Index & 1 and reset on a frame-index gap (ScreenSpaceAmbientOcclusion.cpp:797-800, 1044-1045), SSR radiance / variance history
// (ScreenSpaceReflection.cpp:1045-1046, reset :764-767), the TAA accumulation buffers (TemporalAntiAliasing.cpp:123-143, 272-274).  A drop-in that shards a frame
// across GPUs, re-tiles it, or wants reproducible multi-frame parity runs has to move that state: `export` copies the planes the NEXT frame
// would reproject into caller-owned images together with the index of the frame that wrote them; `import` overwrites them so that the next
// execute with FrameDesc.Index == frame_index + 1 continues the accumulation exactly as if this object had run frame `frame_index` itself.
#include "mifx_objects.h"

using namespace mifx;

namespace
{
uint32_t texel_bytes(uint32_t fmt) { return texel_size(fmt); }

mifx_status copy_plane(mifx_postfx* ctx, const mifx_image2d* dst, const mifx_image2d* src, uint32_t fmt, uint32_t W, uint32_t H, const char* what)
{
    MIFX_REQUIRE(dst != nullptr && src != nullptr && dst->data != nullptr && src->data != nullptr, "%s: null image", what);
    fmt = storage_format(fmt);
    MIFX_REQUIRE(dst->format == fmt && src->format == fmt, "%s: format %u / %u, expected %u", what, dst->format, src->format, fmt);
    MIFX_REQUIRE(dst->width == W && dst->height == H && src->width == W && src->height == H, "%s: %ux%u / %ux%u, the effect is prepared for %ux%u", what, dst->width,
                 dst->height, src->width, src->height, W, H);
    ctx->queued_outside_execute();
    const size_t row = size_t(W) * texel_bytes(fmt);
    MIFX_REQUIRE(dst->pitch_bytes >= row && src->pitch_bytes >= row, "%s: row pitch smaller than a row", what);
    MIFX_HIP_CHECK(hipMemcpy2DAsync(dst->data, dst->pitch_bytes, src->data, src->pitch_bytes, row, H, hipMemcpyDeviceToDevice, ctx->stream));
    return MIFX_OK;
}
} // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------ SSAO: resolved AO (A8) + history length (A5)
mifx_status mifx_ssao_export_history(mifx_ssao* fx, const mifx_image2d* out_ao, const mifx_image2d* out_history_length, uint32_t* out_frame_index)
{
    MIFX_REQUIRE(fx != nullptr && out_frame_index != nullptr, "mifx_ssao_export_history: null argument");
    if (!fx->prepared || fx->last_frame == ~0u || fx->force_reset)
    {
        set_error("mifx_ssao_export_history: no history (nothing executed since the last reset)");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    const int ci = int(fx->last_frame & 1u);
    const mifx_image2d ao = fx->history_ao[ci].desc(), len = fx->history_len[ci].desc();
    MIFX_CHECK(copy_plane(fx->ctx, out_ao, &ao, MIFX_PLANE_AO, fx->w, fx->h, "mifx_ssao_export_history (ao)"));
    MIFX_CHECK(copy_plane(fx->ctx, out_history_length, &len, MIFX_PLANE_HISTORY_LEN, fx->w, fx->h, "mifx_ssao_export_history (history length)"));
    *out_frame_index = fx->last_frame;
    return MIFX_OK;
}

mifx_status mifx_ssao_import_history(mifx_ssao* fx, const mifx_image2d* ao, const mifx_image2d* history_length, uint32_t frame_index)
{
    MIFX_REQUIRE(fx != nullptr && frame_index != ~0u, "mifx_ssao_import_history: bad argument");
    if (!fx->prepared)
    {
        set_error("mifx_ssao_import_history: mifx_ssao_prepare must be called first (the planes take the prepared size)");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    const int ci = int(frame_index & 1u);
    const mifx_image2d dao = fx->history_ao[ci].desc(), dlen = fx->history_len[ci].desc();
    MIFX_CHECK(copy_plane(fx->ctx, &dao, ao, MIFX_PLANE_AO, fx->w, fx->h, "mifx_ssao_import_history (ao)"));
    MIFX_CHECK(copy_plane(fx->ctx, &dlen, history_length, MIFX_PLANE_HISTORY_LEN, fx->w, fx->h, "mifx_ssao_import_history (history length)"));
    fx->last_frame  = frame_index;
    fx->force_reset = false;
    return MIFX_OK;
}

// ------------------------------------------------------------------------------------------------ SSR: accumulated radiance + variance (R6)
mifx_status mifx_ssr_export_history(mifx_ssr* fx, const mifx_image2d* out_radiance, const mifx_image2d* out_variance, uint32_t* out_frame_index)
{
    MIFX_REQUIRE(fx != nullptr && out_frame_index != nullptr, "mifx_ssr_export_history: null argument");
    if (!fx->prepared || fx->last_frame == ~0u)
    {
        set_error("mifx_ssr_export_history: no history (nothing executed since the last reset)");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    const int ci = int(fx->last_frame & 1u);
    const mifx_image2d rad = fx->hist_radiance[ci].desc(), var = fx->hist_variance[ci].desc();
    MIFX_CHECK(copy_plane(fx->ctx, out_radiance, &rad, MIFX_FORMAT_F32X4, fx->w, fx->h, "mifx_ssr_export_history (radiance)"));
    MIFX_CHECK(copy_plane(fx->ctx, out_variance, &var, MIFX_PLANE_VARIANCE, fx->w, fx->h, "mifx_ssr_export_history (variance)"));
    *out_frame_index = fx->last_frame;
    return MIFX_OK;
}

mifx_status mifx_ssr_import_history(mifx_ssr* fx, const mifx_image2d* radiance, const mifx_image2d* variance, uint32_t frame_index)
{
    MIFX_REQUIRE(fx != nullptr && frame_index != ~0u, "mifx_ssr_import_history: bad argument");
    if (!fx->prepared)
    {
        set_error("mifx_ssr_import_history: mifx_ssr_prepare must be called first (the planes take the prepared size)");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    const int ci = int(frame_index & 1u);
    const mifx_image2d drad = fx->hist_radiance[ci].desc(), dvar = fx->hist_variance[ci].desc();
    MIFX_CHECK(copy_plane(fx->ctx, &drad, radiance, MIFX_FORMAT_F32X4, fx->w, fx->h, "mifx_ssr_import_history (radiance)"));
    MIFX_CHECK(copy_plane(fx->ctx, &dvar, variance, MIFX_PLANE_VARIANCE, fx->w, fx->h, "mifx_ssr_import_history (variance)"));
    fx->last_frame = frame_index;
    return MIFX_OK;
}

// ------------------------------------------------------------------------------------------------ TAA: accumulated colour (alpha = accumulated weight)
mifx_status mifx_taa_export_history(mifx_taa* fx, const mifx_image2d* out_color, uint32_t* out_frame_index)
{
    MIFX_REQUIRE(fx != nullptr && out_frame_index != nullptr, "mifx_taa_export_history: null argument");
    if (!fx->prepared || fx->last_frame == ~0u)
    {
        set_error("mifx_taa_export_history: no history (nothing executed since the last reset)");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    const mifx_image2d acc = fx->accum[fx->last_frame & 1u].desc();
    MIFX_CHECK(copy_plane(fx->ctx, out_color, &acc, MIFX_FORMAT_F32X4, fx->w, fx->h, "mifx_taa_export_history"));
    *out_frame_index = fx->last_frame;
    return MIFX_OK;
}

mifx_status mifx_taa_import_history(mifx_taa* fx, const mifx_image2d* color, uint32_t frame_index)
{
    MIFX_REQUIRE(fx != nullptr && frame_index != ~0u, "mifx_taa_import_history: bad argument");
    if (!fx->prepared)
    {
        set_error("mifx_taa_import_history: mifx_taa_prepare must be called first (the planes take the prepared size)");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    const mifx_image2d dst = fx->accum[frame_index & 1u].desc();
    MIFX_CHECK(copy_plane(fx->ctx, &dst, color, MIFX_FORMAT_F32X4, fx->w, fx->h, "mifx_taa_import_history"));
    fx->last_frame = frame_index;
    // "as if this object had run frame_index": an object that has run a frame has created its technique, so the next execute accumulates instead of taking the
    // placeholder copy of a flag set's first frame (mifx_objects.h techniques_created) -- which would overwrite what was just imported.  Which flag sets the exporting
    // object had run is not part of the exported state: all of them count as created.
    fx->techniques_created = 0xFFu;
    fx->technique_ready    = true;
    return MIFX_OK;
}

} // extern "C"
