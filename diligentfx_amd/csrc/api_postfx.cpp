// api_postfx.cpp -- C ABI of PostFXContext (shared per-frame state) and of the stand-alone full-screen passes
// (tone mapping).  Host logic follows PostProcess/Common/src/PostFXContext.cpp:139-338.
#include "mifx_objects.h"

using namespace mifx;

mifx_postfx::~mifx_postfx()
{
    for (hipEvent_t e : timed_events)
        if (e) (void)hipEventDestroy(e);
    if (sobol_dev) (void)hipFree(sobol_dev);
    if (scrambling_dev) (void)hipFree(scrambling_dev);
}

extern "C" {

mifx_status mifx_postfx_set_kernel_timing(mifx_postfx* ctx, const char* kernel_name, uint32_t slots)
{
    MIFX_REQUIRE(ctx != nullptr, "mifx_postfx_set_kernel_timing: null context");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    for (hipEvent_t e : ctx->timed_events)
        if (e) (void)hipEventDestroy(e);
    ctx->timed_events.clear();
    ctx->timed_launches = 0;
    ctx->timed_kernel   = kernel_name ? kernel_name : "";
    if (ctx->timed_kernel.empty() || slots == 0) { ctx->timed_kernel.clear(); return MIFX_OK; }
    ctx->timed_events.assign(size_t(slots) * 2, nullptr);
    for (hipEvent_t& e : ctx->timed_events) MIFX_HIP_CHECK(hipEventCreate(&e));
    return MIFX_OK;
}

mifx_status mifx_postfx_get_kernel_times(mifx_postfx* ctx, float* out_ms, uint32_t capacity, uint32_t* out_count)
{
    MIFX_REQUIRE(ctx != nullptr && out_count != nullptr && (out_ms != nullptr || capacity == 0), "mifx_postfx_get_kernel_times: null argument");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const uint32_t n = ctx->timed_launches < capacity ? ctx->timed_launches : capacity;
    for (uint32_t i = 0; i < n; ++i)
    {
        MIFX_HIP_CHECK(hipEventSynchronize(ctx->timed_events[2 * i + 1]));
        MIFX_HIP_CHECK(hipEventElapsedTime(&out_ms[i], ctx->timed_events[2 * i], ctx->timed_events[2 * i + 1]));
    }
    *out_count = n;
    return MIFX_OK;
}

mifx_status mifx_debug_eval_math(mifx_postfx* ctx, uint32_t op, const float* a, const float* b, float* out, uint64_t n)
{
    MIFX_REQUIRE(ctx != nullptr && a != nullptr && out != nullptr, "mifx_debug_eval_math: null argument");
    MIFX_REQUIRE(op <= MIFX_MATH_POW, "mifx_debug_eval_math: unknown operation %u", op);
    MIFX_REQUIRE(b != nullptr || (op != MIFX_MATH_FDIV && op != MIFX_MATH_POW), "mifx_debug_eval_math: binary operation needs b");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return mifx::launch_eval_math(ctx->stream, op, a, b, out, n);
}

mifx_status mifx_debug_stream_copy(mifx_postfx* ctx, const void* src, void* dst, uint64_t bytes)
{
    MIFX_REQUIRE(ctx != nullptr && src != nullptr && dst != nullptr && (bytes & 15u) == 0 && ((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0,
                 "mifx_debug_stream_copy: null or unaligned argument (16-byte granularity)");
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return mifx::launch_stream_copy(ctx->stream, src, dst, bytes);
}

mifx_status mifx_postfx_create(const mifx_device_desc* dev, const mifx_postfx_create_info* info, mifx_postfx** out)
{
    MIFX_REQUIRE(dev != nullptr && out != nullptr, "mifx_postfx_create: dev and out must not be null");
    *out = nullptr;
    int count = 0;
    MIFX_HIP_CHECK(hipGetDeviceCount(&count));
    MIFX_REQUIRE(dev->device >= 0 && dev->device < count, "mifx_postfx_create: device %d out of range (%d devices)", dev->device, count);
    MIFX_HIP_CHECK(hipSetDevice(dev->device));
    mifx_postfx* ctx = new mifx_postfx();
    ctx->device      = dev->device;
    ctx->stream      = static_cast<hipStream_t>(dev->hip_stream);
    if (info && info->sobol_256d && info->scrambling_tile)
    {
        // PostFXContext.cpp:152-190: Sobol 256x1 R8_UINT, scrambling tile 512x256 R8_UINT (= 128*128*8 bytes)
        hipError_t e = hipMalloc(&ctx->sobol_dev, 256);
        if (e == hipSuccess) e = hipMalloc(&ctx->scrambling_dev, 128 * 128 * 8);
        if (e == hipSuccess) e = hipMemcpy(ctx->sobol_dev, info->sobol_256d, 256, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(ctx->scrambling_dev, info->scrambling_tile, 128 * 128 * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess)
        {
            set_error("mifx_postfx_create: uploading blue-noise tables failed: %s", hipGetErrorString(e));
            delete ctx;
            return MIFX_ERR_HIP;
        }
    }
    *out = ctx;
    return MIFX_OK;
}

void mifx_postfx_destroy(mifx_postfx* ctx) { delete ctx; }

mifx_status mifx_postfx_set_stream(mifx_postfx* ctx, void* hip_stream)
{
    MIFX_REQUIRE(ctx != nullptr, "mifx_postfx_set_stream: ctx must not be null");
    if (ctx->stream != static_cast<hipStream_t>(hip_stream)) ctx->queued_outside_execute(); // (a chain's lanes are ordered behind the new stream once)
    ctx->stream = static_cast<hipStream_t>(hip_stream);
    return MIFX_OK;
}

// PostFXContext::PrepareResources (PostFXContext.cpp:241-285): (re)allocate on size / flag change only.
mifx_status mifx_postfx_prepare(mifx_postfx* ctx, const mifx_frame_desc* frame, uint32_t feature_flags)
{
    MIFX_REQUIRE(ctx != nullptr && frame != nullptr, "mifx_postfx_prepare: ctx and frame must not be null");
    MIFX_REQUIRE(frame->Width > 0 && frame->Height > 0, "mifx_postfx_prepare: empty frame %ux%u", frame->Width, frame->Height);
    MIFX_REQUIRE((feature_flags & ~7u) == 0, "mifx_postfx_prepare: unknown feature flags 0x%x", feature_flags);
    // FEATURE_FLAG_TEMPORAL_UPSCALING (PostFXContext.hpp:56): the effects behind the up-scaler (Bloom, Bloom.cpp:84-85) take FrameDesc.OutputWidth x OutputHeight
    MIFX_REQUIRE(!(feature_flags & MIFX_POSTFX_FEATURE_FLAG_TEMPORAL_UPSCALING) || (frame->OutputWidth > 0 && frame->OutputHeight > 0),
                 "mifx_postfx_prepare: FEATURE_FLAG_TEMPORAL_UPSCALING needs FrameDesc.OutputWidth / OutputHeight (got %ux%u)", frame->OutputWidth, frame->OutputHeight);
    // FEATURE_FLAG_HALF_PRECISION_DEPTH selects R16_UNORM for the reprojected / previous depth (PostFXContext.cpp:259,270) -- when the targets are created, i.e. on a
    // change of the frame size only (:246-247).  fp32 build: full precision like every plane; native-storage build: see mifx_postfx::depth16.
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const bool recreated = ctx->reproj_depth.data == nullptr || ctx->reproj_depth.w != frame->Width || ctx->reproj_depth.h != frame->Height;
    ctx->frame = *frame;
    ctx->flags = feature_flags;
    MIFX_CHECK(ctx->reproj_depth.alloc(frame->Width, frame->Height, MIFX_FORMAT_F32));
    if (recreated)
    {
        ctx->depth16 = mifx_storage_mode() == MIFX_STORAGE_RGBA16F && (feature_flags & MIFX_POSTFX_FEATURE_FLAG_HALF_PRECISION_DEPTH) != 0;
        if (ctx->depth16) MIFX_CHECK(ctx->prev_depth16.alloc(frame->Width, frame->Height, MIFX_FORMAT_F32));
        else ctx->prev_depth16.release();
    }
    MIFX_CHECK(ctx->closest_motion.alloc(frame->Width, frame->Height, MIFX_PLANE_CLOSEST_MOTION));
    MIFX_CHECK(ctx->noise_xy.alloc(128, 128, MIFX_FORMAT_F32X2));
    MIFX_CHECK(ctx->noise_zw.alloc(128, 128, MIFX_FORMAT_F32X2));
    ctx->prepared = true;
    ctx->executed = false;
    return MIFX_OK;
}

// PostFXContext::Execute (PostFXContext.cpp:287-338): blue noise (C1), reprojected depth (C2), closest motion (C3),
// previous depth (C4; a pure copy in the reference, an alias of the borrowed plane here).
mifx_status mifx_postfx_execute(mifx_postfx* ctx, const mifx_postfx_render_attribs* a)
{
    MIFX_REQUIRE(ctx != nullptr && a != nullptr, "mifx_postfx_execute: null argument");
    if (!ctx->prepared)
    {
        set_error("mifx_postfx_execute: mifx_postfx_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_REQUIRE(a->curr_camera && a->prev_camera, "mifx_postfx_execute: cameras must not be null");
    const uint32_t W = ctx->frame.Width, H = ctx->frame.Height;
    MIFX_RANGE("PreparePostFX");
    Img depth, prev_depth, motion;
    MIFX_CHECK(to_img_wh(a->curr_depth, MIFX_FORMAT_F32, W, H, "curr_depth", depth));
    MIFX_CHECK(to_img_wh(a->prev_depth, MIFX_FORMAT_F32, W, H, "prev_depth", prev_depth));
    MIFX_CHECK(to_img_wh(a->motion, MIFX_FORMAT_F32X2, W, H, "motion", motion));
    ctx->curr_cam   = *a->curr_camera;
    ctx->prev_cam   = *a->prev_camera;
    ctx->prev_depth = *a->prev_depth;
    ctx->prep_rows = ctx->needed_rows(int(depth.h)); // C2 / C3 read only the frame inputs: any row window is exact
    const bool rev = (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0;
    MifxKernelTimer timer(ctx, "postfx_prep_kernel");
    // (C1, the blue noise of the frame, is written by extra workgroups of the same launch: prep.hip)
    MIFX_CHECK(launch_postfx_prep(ctx->stream, win(depth, ctx->prep_rows), motion, ctx->reproj_depth.view(), ctx->closest_motion.view(), make_camk(ctx->curr_cam, rev),
                                  make_camk(ctx->prev_cam, rev), static_cast<const uint8_t*>(ctx->sobol_dev), static_cast<const uint8_t*>(ctx->scrambling_dev), ctx->noise_xy.view(),
                                  ctx->noise_zw.view(), ctx->frame.Index, ctx->depth16));
    if (ctx->depth16) // ComputePreviousDepth: the copy into the R16_UNORM target (PostFXContext.cpp:325-337); the effects read it through mifx_postfx::prev_depth
    {
        MIFX_CHECK(launch_depth16_copy(ctx->stream, prev_depth, ctx->prev_depth16.view()));
        ctx->prev_depth = ctx->prev_depth16.desc();
    }
    ctx->executed = true;
    return MIFX_OK;
}

// ------------------------------------------------------------------------------------------------ PostFXContext's public texture helpers (PostFXContext.hpp:114, 168-172)
// GetSupportedFeatures: what the reference derives from the device (PostFXContext.cpp:145-148) and its effects branch on.  Here every pass addresses every mip level of
// every plane directly and takes its frame / instance index as a kernel argument, so the four capabilities hold by construction.
mifx_status mifx_postfx_get_supported_features(mifx_postfx* ctx, mifx_postfx_supported_features* out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_postfx_get_supported_features: null argument");
    out->TransitionSubresources = out->TextureSubresourceViews = out->CopyDepthToColor = out->ShaderBaseVertexOffset = 1;
    return MIFX_OK;
}

static mifx_status plane_channels(const mifx_image2d* im, const char* what, int& channels, bool& halves)
{
    MIFX_REQUIRE(im != nullptr && im->data != nullptr && im->width > 0 && im->height > 0, "%s: null or empty image", what);
    switch (im->format)
    {
        case MIFX_FORMAT_F32: channels = 1; halves = false; break;
        case MIFX_FORMAT_F32X2: channels = 2; halves = false; break;
        case MIFX_FORMAT_F32X4: channels = 4; halves = false; break;
        case MIFX_FORMAT_F16: channels = 1; halves = true; break;
        case MIFX_FORMAT_F16X2: channels = 2; halves = true; break;
        case MIFX_FORMAT_F16X4: channels = 4; halves = true; break;
        default: set_error("%s: format %u is not a float plane", what, im->format); return MIFX_ERR_INVALID_ARG;
    }
    MIFX_REQUIRE(size_t(im->pitch_bytes) >= size_t(im->width) * size_t(channels) * (halves ? 2u : 4u), "%s: row pitch smaller than a row", what);
    return MIFX_OK;
}

// PostFXContext::ClearRenderTarget (PostFXContext.cpp:347-353): the whole target := ClearColor (one value per channel)
mifx_status mifx_postfx_clear_render_target(mifx_postfx* ctx, const mifx_image2d* target, const float clear_color[4])
{
    MIFX_REQUIRE(ctx != nullptr && clear_color != nullptr, "mifx_postfx_clear_render_target: null argument");
    int  channels = 0;
    bool halves   = false;
    MIFX_CHECK(plane_channels(target, "mifx_postfx_clear_render_target", channels, halves));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const Img im{static_cast<unsigned char*>(target->data), int(target->width), int(target->height), int(target->pitch_bytes), 0, 0};
    return launch_clear_texels(ctx->stream, im, channels, halves, clear_color);
}

// PostFXContext::CopyTextureDepth / CopyTextureColor (PostFXContext.cpp:355-438): a full-screen draw that samples the source at the texel centres of the target with a
// point (depth) / linear (colour) CLAMP sampler.  Every caller in the reference copies between targets of one size (mip 0 of a pyramid, TAA's placeholder frame, the
// previous depth), where both samplers return the texel itself: the entry points take equal sizes and formats and copy the rows.
static mifx_status copy_texture(mifx_postfx* ctx, const mifx_image2d* src, const mifx_image2d* dst, const char* what)
{
    MIFX_REQUIRE(ctx != nullptr, "%s: null context", what);
    int  cs = 0, cd = 0;
    bool hs = false, hd = false;
    MIFX_CHECK(plane_channels(src, what, cs, hs));
    MIFX_CHECK(plane_channels(dst, what, cd, hd));
    MIFX_REQUIRE(src->format == dst->format && src->width == dst->width && src->height == dst->height, "%s: %ux%u format %u -> %ux%u format %u (the copy is between targets of one size and format)", what,
                 src->width, src->height, src->format, dst->width, dst->height, dst->format);
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    MIFX_HIP_CHECK(hipMemcpy2DAsync(dst->data, dst->pitch_bytes, src->data, src->pitch_bytes, size_t(src->width) * size_t(cs) * (hs ? 2u : 4u), src->height, hipMemcpyDeviceToDevice, ctx->stream));
    return MIFX_OK;
}
mifx_status mifx_postfx_copy_texture_depth(mifx_postfx* ctx, const mifx_image2d* src, const mifx_image2d* dst) { return copy_texture(ctx, src, dst, "mifx_postfx_copy_texture_depth"); }
mifx_status mifx_postfx_copy_texture_color(mifx_postfx* ctx, const mifx_image2d* src, const mifx_image2d* dst) { return copy_texture(ctx, src, dst, "mifx_postfx_copy_texture_color"); }

static mifx_status get_plane(mifx_postfx* ctx, const Plane& p, mifx_image2d* out, const char* what)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "%s: null argument", what);
    if (!ctx->prepared || !p.data)
    {
        set_error("%s: resources are not prepared", what);
        return MIFX_ERR_INVALID_OP;
    }
    *out = p.desc();
    return MIFX_OK;
}
mifx_status mifx_postfx_get_reprojected_depth(mifx_postfx* ctx, mifx_image2d* out) { return get_plane(ctx, ctx->reproj_depth, out, "mifx_postfx_get_reprojected_depth"); }
mifx_status mifx_postfx_get_closest_motion(mifx_postfx* ctx, mifx_image2d* out) { return get_plane(ctx, ctx->closest_motion, out, "mifx_postfx_get_closest_motion"); }
mifx_status mifx_postfx_get_previous_depth(mifx_postfx* ctx, mifx_image2d* out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_postfx_get_previous_depth: null argument");
    if (!ctx->executed)
    {
        set_error("mifx_postfx_get_previous_depth: mifx_postfx_execute has not run for this frame");
        return MIFX_ERR_INVALID_OP;
    }
    *out = ctx->prev_depth;
    return MIFX_OK;
}
mifx_status mifx_postfx_get_blue_noise(mifx_postfx* ctx, int32_t dimension, mifx_image2d* out)
{
    MIFX_REQUIRE(ctx != nullptr, "mifx_postfx_get_blue_noise: null ctx");
    MIFX_REQUIRE(dimension == 0 || dimension == 1, "mifx_postfx_get_blue_noise: dimension must be 0 (XY) or 1 (ZW)");
    if (!ctx->sobol_dev)
    {
        set_error("mifx_postfx_get_blue_noise: no blue-noise tables were supplied to mifx_postfx_create");
        return MIFX_ERR_INVALID_OP;
    }
    return get_plane(ctx, dimension == 0 ? ctx->noise_xy : ctx->noise_zw, out, "mifx_postfx_get_blue_noise");
}

// ------------------------------------------------------------------------------------------------ tone mapping
mifx_status mifx_tonemap_execute(mifx_postfx* ctx, const mifx_image2d* hdr_in, const mifx_image2d* ldr_out, const mifx_tone_mapping_attribs* attribs,
                                 float ave_log_lum, uint32_t flags)
{
    MIFX_REQUIRE(ctx != nullptr && attribs != nullptr, "mifx_tonemap_execute: null argument");
    MIFX_REQUIRE(attribs->iToneMappingMode >= 0 && attribs->iToneMappingMode <= MIFX_TONE_MAPPING_MODE_COMMERCE, "mifx_tonemap_execute: unknown tone mapping mode %d",
                 attribs->iToneMappingMode);
    MIFX_REQUIRE((flags & ~uint32_t(MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB)) == 0, "mifx_tonemap_execute: unknown flags 0x%x", flags);
    Img  in, out;
    bool packed = false; // (native-storage build: Bloom's own R11G11B10_FLOAT output plane is accepted beside the RGBA16_FLOAT frame)
    MIFX_CHECK(to_img_hdr(hdr_in, "hdr_in", in, packed));
    MIFX_CHECK(to_img_wh(ldr_out, MIFX_FORMAT_F32X4, hdr_in->width, hdr_in->height, "ldr_out", out));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    MifxKernelTimer timer(ctx, "tonemap_kernel");
    return launch_tonemap(ctx->stream, in, win(out, ctx->needed_rows(out.h)), *attribs, ave_log_lum, flags, nullptr, packed);
}

mifx_status mifx_tonemap_execute_native(mifx_postfx* ctx, const mifx_image2d* hdr_in, const mifx_native_image* ldr_out, const mifx_tone_mapping_attribs* attribs, float ave_log_lum,
                                        uint32_t flags)
{
    MIFX_REQUIRE(ctx != nullptr && attribs != nullptr && ldr_out != nullptr, "mifx_tonemap_execute_native: null argument");
    MIFX_REQUIRE(attribs->iToneMappingMode >= 0 && attribs->iToneMappingMode <= MIFX_TONE_MAPPING_MODE_COMMERCE, "mifx_tonemap_execute_native: unknown tone mapping mode %d",
                 attribs->iToneMappingMode);
    MIFX_REQUIRE((flags & ~uint32_t(MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB)) == 0, "mifx_tonemap_execute_native: unknown flags 0x%x", flags);
    MIFX_REQUIRE(ctx->band.empty(), "mifx_tonemap_execute_native: not available with a row band");
    Img  in;
    bool packed = false;
    MIFX_CHECK(to_img_hdr(hdr_in, "hdr_in", in, packed));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    return launch_tonemap_native(ctx->stream, in, ldr_out, *attribs, ave_log_lum, flags, nullptr, packed);
}

// Components/src/ToneMapping.cpp:43-83 (ReverseExpToneMap): inverse of the EXP operator for a given LDR colour.
mifx_status mifx_reverse_exp_tone_map(const float ldr[3], float middle_gray, float ave_log_lum, float out_hdr[3])
{
    MIFX_REQUIRE(ldr != nullptr && out_hdr != nullptr, "mifx_reverse_exp_tone_map: null argument");
    const float lum = 0.212671f * ldr[0] + 0.715160f * ldr[1] + 0.072169f * ldr[2];
    if (lum == 0.0f)
    {
        out_hdr[0] = out_hdr[1] = out_hdr[2] = 0.0f;
        return MIFX_OK;
    }
    // grey-scale / saturation-1 inversion of the EXP operator: InitialLum = -log(1 - ToneMappedLum) / LumScale
    const float lum_scale = middle_gray / ave_log_lum;
    const float t         = 1.0f - lum;
    const float initial   = -logf(t > 0.01f ? t : 0.01f) / lum_scale;
    out_hdr[0] = ldr[0] * initial / lum; out_hdr[1] = ldr[1] * initial / lum; out_hdr[2] = ldr[2] * initial / lum;
    return MIFX_OK;
}

} // extern "C"
