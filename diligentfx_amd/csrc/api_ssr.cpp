// api_ssr.cpp -- C ABI + host sequencing of ScreenSpaceReflection
// (PostProcess/ScreenSpaceReflection/src/ScreenSpaceReflection.cpp: PrepareResources :67-298, Execute :300-341, Compute* :777-1104).
#include "mifx_objects.h"
#include <cmath>
#include <cstdlib>

using namespace mifx;

// The depth hierarchy of a W x H frame: one allocation, level k a pitched view into it (level 0 = the copy of the depth buffer).  Also used by the chain for the second
// copy its pipelined mode keeps (api_chain.cpp).
mifx_status mifx::ssr_alloc_hiz(uint32_t W, uint32_t H, Plane* hiz, DeviceScratch& slab)
{
    size_t total = 0, off[mifx_ssr::kMips];
    uint32_t lw[mifx_ssr::kMips], lh[mifx_ssr::kMips], lp[mifx_ssr::kMips];
    for (int k = 0; k < mifx_ssr::kMips; ++k)
    {
        lw[k] = (W >> k) ? (W >> k) : 1u; lh[k] = (H >> k) ? (H >> k) : 1u;
        lp[k] = ((lw[k] * 4u + 255u) / 256u) * 256u;
        off[k] = total;
        total += size_t(lp[k]) * lh[k];
    }
    MIFX_REQUIRE(total < (size_t(1) << 32), "mifx_ssr_prepare: depth hierarchy of %ux%u exceeds the 32-bit offset range", W, H);
    for (int k = 0; k < mifx_ssr::kMips; ++k) hiz[k].release();
    MIFX_CHECK(slab.reserve(total));
    for (int k = 0; k < mifx_ssr::kMips; ++k) hiz[k].attach(static_cast<unsigned char*>(slab.data) + off[k], lw[k], lh[k], lp[k], MIFX_FORMAT_F32);
    return MIFX_OK;
}

extern "C" {

mifx_status mifx_ssr_create(mifx_postfx* ctx, mifx_ssr** out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_ssr_create: null argument");
    *out        = new mifx_ssr();
    (*out)->ctx = ctx;
    if (const char* e = std::getenv("MIFX_SSR_DIRECT_LEVEL0")) (*out)->direct_level0 = std::atoi(e) < 0 ? -1 : std::atoi(e) > 0 ? 1 : 0; // (A/B runs; default -1: row bands only)
    return MIFX_OK;
}
void mifx_ssr_destroy(mifx_ssr* fx) { delete fx; }

static mifx_status clear_history(mifx_ssr* fx)
{
    // radiance / variance history and the output are cleared to 0 when (re)created (.cpp:262-264, 278-280, 293-295)
    fx->ctx->queued_outside_execute();
    for (int i = 0; i < 2; ++i)
    {
        MIFX_CHECK(fx->hist_radiance[i].fill(fx->ctx->stream, 0.0f));
        MIFX_CHECK(fx->hist_variance[i].fill(fx->ctx->stream, 0.0f));
    }
    // R5's targets are written under the reflection mask only and read beside its edge: zero like a new render target (the reference never clears them)
    MIFX_CHECK(fx->res_radiance.fill(fx->ctx->stream, 0.0f));
    MIFX_CHECK(fx->res_variance.fill(fx->ctx->stream, 0.0f));
    MIFX_CHECK(fx->res_depth.fill(fx->ctx->stream, 0.0f));
    return fx->output.fill(fx->ctx->stream, 0.0f);
}

mifx_status mifx_ssr_prepare(mifx_ssr* fx, mifx_postfx* ctx, uint32_t feature_flags)
{
    MIFX_REQUIRE(fx != nullptr && ctx != nullptr, "mifx_ssr_prepare: null argument");
    if (!ctx->prepared)
    {
        set_error("mifx_ssr_prepare: mifx_postfx_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_REQUIRE((feature_flags & ~3u) == 0, "mifx_ssr_prepare: unknown feature flags 0x%x", feature_flags);
    const bool half = (feature_flags & MIFX_SSR_FEATURE_FLAG_HALF_RESOLUTION) != 0;
    MIFX_REQUIRE(!half || (ctx->frame.Width >= 4 && ctx->frame.Height >= 4), "mifx_ssr_prepare: frame too small for the half-resolution ray pass");
    fx->ctx = ctx;
    const uint32_t W = ctx->frame.Width, H = ctx->frame.Height;
    // The targets are (re-)created -- the histories cleared -- on a change of the frame size or of FEATURE_FLAG_HALF_RESOLUTION; FEATURE_FLAG_PREVIOUS_FRAME only selects another
    // permutation of the ray march and keeps everything (ScreenSpaceReflection.cpp:72-85; found by executing that file, oracle/refhost: rounds 1-3 cleared the history here too)
    if (fx->prepared && fx->w == W && fx->h == H && ((fx->flags ^ feature_flags) & MIFX_SSR_FEATURE_FLAG_HALF_RESOLUTION) == 0)
    {
        fx->flags = feature_flags;
        return MIFX_OK;
    }
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    fx->prepared = false; // ready again only when every plane of the new size exists (see mifx_ssao_prepare)
    MIFX_CHECK(mifx::ssr_alloc_hiz(W, H, fx->hiz, fx->hiz_slab));
    MIFX_CHECK(fx->roughness.alloc(W, H, MIFX_PLANE_ROUGHNESS));
    MIFX_CHECK(fx->mask.alloc(W, H, MIFX_PLANE_MASK));
    // FEATURE_FLAG_HALF_RESOLUTION: the ray textures and their mask are (W / 2) x (H / 2) (ScreenSpaceReflection.cpp:181-190, 201-213)
    const uint32_t RW = half ? W / 2u : W, RH = half ? H / 2u : H;
    MIFX_CHECK(fx->ray_radiance.alloc(RW, RH, MIFX_FORMAT_F32X4));
    MIFX_CHECK(fx->ray_dir_pdf.alloc(RW, RH, MIFX_FORMAT_F32X4));
    if (half) MIFX_CHECK(fx->mask_half.alloc(RW, RH, MIFX_PLANE_MASK));
    else fx->mask_half.release();
    MIFX_CHECK(fx->res_radiance.alloc(W, H, MIFX_FORMAT_F32X4));
    MIFX_CHECK(fx->res_variance.alloc(W, H, MIFX_PLANE_VARIANCE));
    MIFX_CHECK(fx->res_depth.alloc(W, H, MIFX_PLANE_VARIANCE));
    for (int i = 0; i < 2; ++i)
    {
        MIFX_CHECK(fx->hist_radiance[i].alloc(W, H, MIFX_FORMAT_F32X4));
        MIFX_CHECK(fx->hist_variance[i].alloc(W, H, MIFX_PLANE_VARIANCE));
    }
    MIFX_CHECK(fx->output.alloc(W, H, MIFX_FORMAT_F32X4));
    fx->w = W; fx->h = H; fx->flags = feature_flags;
    MIFX_CHECK(clear_history(fx));
    fx->last_frame = ~0u;
    fx->prepared   = true;
    return MIFX_OK;
}

mifx_status mifx_ssr_reset_history(mifx_ssr* fx)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_ssr_reset_history: null argument");
    fx->last_frame = ~0u;
    return fx->prepared ? clear_history(fx) : MIFX_OK;
}

mifx_status mifx_ssr_execute(mifx_ssr* fx, const mifx_ssr_render_attribs* ra)
{
    MIFX_REQUIRE(fx != nullptr && ra != nullptr && ra->attribs != nullptr, "mifx_ssr_execute: null argument");
    const hipStream_t hizStream = fx->hiz_stream; // (a per-frame request: taken and cleared before anything can return)
    const hipEvent_t  hizDone   = fx->hiz_done;
    fx->hiz_stream = nullptr;
    fx->hiz_done   = nullptr;
    mifx_postfx* ctx = ra->postfx ? ra->postfx : fx->ctx;
    if (!fx->prepared || !ctx || !ctx->executed)
    {
        set_error("mifx_ssr_execute: call mifx_ssr_prepare and mifx_postfx_execute for this frame first");
        return MIFX_ERR_INVALID_OP;
    }
    if (!ctx->sobol_dev)
    {
        set_error("mifx_ssr_execute: the PostFX context has no blue-noise tables");
        return MIFX_ERR_INVALID_OP;
    }
    const uint32_t W = fx->w, H = fx->h;
    const mifx_ssr_attribs& a = *ra->attribs;
    MIFX_REQUIRE(a.RoughnessChannel <= 3u, "mifx_ssr_execute: RoughnessChannel %u out of range", a.RoughnessChannel);
    MIFX_REQUIRE(a.MostDetailedMip <= 6u, "mifx_ssr_execute: MostDetailedMip %u exceeds SSR_DEPTH_HIERARCHY_MAX_MIP", a.MostDetailedMip);
    Img color, depth, normal, material, motion, prevDepth;
    MIFX_CHECK(to_img_wh(ra->color, MIFX_FORMAT_F32X4, W, H, "color", color));
    MIFX_CHECK(to_img_wh(ra->depth, MIFX_FORMAT_F32, W, H, "depth", depth));
    MIFX_CHECK(to_img_wh(ra->normal, MIFX_FORMAT_F32X4, W, H, "normal", normal));
    MIFX_CHECK(to_img_wh(ra->material, MIFX_FORMAT_F32X4, W, H, "material", material));
    MIFX_CHECK(to_img_wh(ra->motion, MIFX_FORMAT_F32X2, W, H, "motion", motion));
    MIFX_CHECK(to_img_wh(&ctx->prev_depth, MIFX_FORMAT_F32, W, H, "previous depth", prevDepth));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const uint32_t idx = ctx->frame.Index;
    fx->last_frame = idx;
    MIFX_RANGE("ScreenSpaceReflection");
    const int  ci = int(idx & 1u), pi = int((idx + 1u) & 1u); // .cpp:1044-1046
    const bool rev = (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0; // SSR_OPTION_INVERTED_DEPTH (ScreenSpaceReflection.cpp:73)
    const CamK cur = make_camk(ctx->curr_cam, rev), prev = make_camk(ctx->prev_cam, rev);

    // R1: closest-depth pyramid (mip 0 = the depth itself, a copy in the reference :789-806)
    Pyr hiz{};
    hiz.levels = mifx_ssr::kMips;
    hiz.l[0]   = depth;
    for (int k = 1; k < mifx_ssr::kMips; ++k) hiz.l[k] = fx->hiz[k].view();
    // Level 0 of the hierarchy is the depth buffer itself.  The reference copies it into mip 0 of its texture (:789-806), and so does the unsharded frame here (the march
    // then addresses all levels through one descriptor).  A row band of a sharded frame builds the WHOLE hierarchy on every rank -- a ray ends anywhere -- and the copy is
    // 86 % of that pass' bytes (57 of 67 us per rank at 7680x4320): there the march reads level 0 where it lies, through a second descriptor (ssr_trace.hip DIRECT0;
    // the same values, one v_cmp and a few scalar instructions more per tap).  direct_level0: -1 = row bands only (default), 0 / 1 = never / always (MIFX_SSR_DIRECT_LEVEL0).
    const size_t depthBytes = size_t(depth.pitch) * size_t(depth.h);
    const bool   direct0    = (fx->direct_level0 > 0 || (fx->direct_level0 < 0 && !ctx->band.empty())) && depthBytes < (size_t(1) << 31) && depth.pitch < (1 << 24) && depth.w < (1 << 24) &&
                              depth.h < (1 << 24);
    const Img level0Copy = direct0 ? Img{} : fx->hiz[0].view();
    if (hizStream != nullptr && hizDone != nullptr)
    {
        MIFX_CHECK(launch_ssr_hiz_pyramid(hizStream, hiz, level0Copy, rev));
        MIFX_HIP_CHECK(hipEventRecord(hizDone, hizStream));
        MIFX_HIP_CHECK(hipStreamWaitEvent(s, hizDone, 0));
    }
    else
        MIFX_CHECK(launch_ssr_hiz_pyramid(s, hiz, level0Copy, rev));
    HizSlab slab{};
    slab.base   = static_cast<const unsigned char*>(fx->hiz_slab.data);
    slab.levels = mifx_ssr::kMips;
    slab.bytes  = uint32_t(fx->hiz_slab.bytes);
    for (int k = 0; k < mifx_ssr::kMips; ++k)
    {
        slab.offset[k] = uint32_t(static_cast<const unsigned char*>(fx->hiz[k].data) - slab.base);
        slab.pitch[k] = fx->hiz[k].pitch; slab.w[k] = fx->hiz[k].w; slab.h[k] = fx->hiz[k].h;
    }
    if (direct0)
    {
        slab.base0    = depth.p;
        slab.bytes0   = uint32_t(depthBytes);
        slab.offset[0] = 0u;
        slab.pitch[0]  = uint32_t(depth.pitch);
    }
    // Row windows (mifx_rows.h), from the rows of the output its consumers need back to the ray march; whole frame by default.
    //   R7 filters +-2 texels and takes quad derivatives (+-1); R6 reads the 3x3 neighbourhood of the resolved radiance (its history taps are
    //   covered by the halo exchange); R5 gathers 8 Poisson taps of radius <= SpatialReconstructionRadius (+1 for the truncation); R4 reads
    //   the whole depth hierarchy, normals and scene colour (all-gathered), R2 is pointwise.
    const int  iH = int(fx->h);
    const Rows w7 = ctx->needed_rows(iH);
    const Rows w6 = rows_expand(w7, 3, iH);
    const Rows w5 = rows_expand(w6, 1, iH);
    const bool half = (fx->flags & MIFX_SSR_FEATURE_FLAG_HALF_RESOLUTION) != 0;
    const Rows w4 = mifx_ssr::march_rows(a, w7, iH, half); // full resolution: rows_expand(w5, radius + 1)
    const Rows h4 = mifx_ssr::half_rows(a, w5, iH);        // half resolution: the rows of the half-size ray textures R5 reads
    MIFX_REQUIRE(ctx->prep_rows.empty() || rows_contain(ctx->prep_rows, w4), "mifx_ssr_execute: PostFX prep covered rows [%d, %d), needed [%d, %d)", ctx->prep_rows.b,
                 ctx->prep_rows.e, w4.b, w4.e);
    // R2
    if (fx->mask_provided_for != idx) // (the chain's shade kernel writes both planes as a by-product: mifx_chain_execute)
    {
        MifxKernelTimer timer(ctx, "ssr_mask_roughness_kernel");
        MIFX_CHECK(launch_ssr_mask_roughness(s, material, depth, fx->roughness.view(), win(fx->mask.view(), w4), a, rev));
    }
    fx->mask_provided_for = ~0u;
    // R3 (half resolution, :934-961): mask of the half-size ray pass
    if (half) MIFX_CHECK(launch_ssr_downsampled_mask(s, fx->roughness.view(), depth, win(fx->mask_half.view(), h4), a, rev));
    // R4 (under the half-size mask in half-resolution mode, :988)
    {
        MifxKernelTimer timer(ctx, "ssr_intersection_kernel");
        Img coords{};
        if (fx->after_trace)
        {
            MIFX_CHECK(fx->hit_coords.alloc(fx->ray_radiance.w, fx->ray_radiance.h, MIFX_FORMAT_F32));
            coords = fx->hit_coords.view();
        }
        MIFX_CHECK(launch_ssr_intersection(s, color, normal, fx->roughness.view(), ctx->noise_xy.view(), slab, half ? fx->mask_half.view() : fx->mask.view(), motion,
                                           win(fx->ray_radiance.view(), half ? h4 : w4), fx->ray_dir_pdf.view(), cur, a,
                                           (fx->flags & MIFX_SSR_FEATURE_FLAG_PREVIOUS_FRAME) != 0, half, coords, fx->hit_local_rows.b, fx->hit_local_rows.e));
    }
    if (fx->after_trace)
    {
        auto hook = std::move(fx->after_trace);
        fx->after_trace    = nullptr;
        fx->hit_local_rows = Rows{0, 0};
        MIFX_CHECK(hook(win(fx->ray_radiance.view(), half ? h4 : w4), fx->hit_coords.view()));
    }
    // R5
    {
        MifxKernelTimer timer(ctx, "ssr_spatial_kernel");
        MIFX_CHECK(launch_ssr_spatial(s, fx->roughness.view(), normal, depth, fx->ray_dir_pdf.view(), fx->ray_radiance.view(), fx->mask.view(), win(fx->res_radiance.view(), w5),
                                      fx->res_variance.view(), fx->res_depth.view(), cur, a, half));
    }
    // R6
    {
        MifxKernelTimer timer(ctx, "ssr_temporal_kernel");
        MIFX_CHECK(launch_ssr_temporal(s, motion, fx->res_depth.view(), ctx->reproj_depth.view(), fx->res_radiance.view(), fx->res_variance.view(), prevDepth,
                                       fx->hist_radiance[pi].view(), fx->hist_variance[pi].view(), fx->mask.view(), win(fx->hist_radiance[ci].view(), w6), fx->hist_variance[ci].view(), cur,
                                       prev, a));
    }
    // R7 (now, or inside the chain's composite: mifx_objects.h `defer_cleanup`)
    fx->cleanup_in     = SsrCleanupIn{depth, fx->roughness.view(), fx->hist_radiance[ci].view(), fx->hist_variance[ci].view(), fx->mask.view(), a.RoughnessThreshold,
                                      a.BilateralCleanupSpatialSigmaFactor, a.AlphaInterpolation, rev ? 1 : 0};
    fx->cleanup_normal = normal;
    fx->cleanup_cam    = cur;
    fx->cleanup_rows   = w7;
    fx->cleanup_pending = true;
    if (!fx->defer_cleanup) MIFX_CHECK(fx->run_cleanup());
    fx->defer_cleanup = false; // (a per-frame request)
    return MIFX_OK;
}

} // extern "C"
mifx_status mifx_ssr::run_cleanup()
{
    if (!cleanup_pending) return MIFX_OK;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    MifxKernelTimer t7(ctx, "ssr_bilateral_kernel");
    MIFX_CHECK(launch_ssr_bilateral(ctx->stream, cleanup_normal, cleanup_in, win(output.view(), cleanup_rows), cleanup_cam));
    cleanup_pending = false;
    return MIFX_OK;
}
extern "C" {

mifx_status mifx_ssr_get_output(mifx_ssr* fx, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && out != nullptr, "mifx_ssr_get_output: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_ssr_get_output: resources are not prepared");
        return MIFX_ERR_INVALID_OP;
    }
    if (fx->cleanup_pending)
    {
        // (the pass reads the caller's depth and normal planes, which were borrowed for the duration of the execute only: it cannot be run from here)
        set_error("mifx_ssr_get_output: the chain evaluated this frame's bilateral cleanup inside its composite and did not write the output plane; "
                  "mifx_ssr_run_deferred_cleanup(depth, normal) produces it, or switch MIFX_CHAIN_FUSE_SSR_CLEANUP_INTO_COMPOSITE off");
        return MIFX_ERR_INVALID_OP;
    }
    *out = fx->output.desc();
    return MIFX_OK;
}

mifx_status mifx_ssr_run_deferred_cleanup(mifx_ssr* fx, const mifx_image2d* depth, const mifx_image2d* normal)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_ssr_run_deferred_cleanup: null argument");
    if (!fx->prepared || !fx->cleanup_pending) return MIFX_OK; // nothing deferred: the output plane is current
    MIFX_CHECK(to_img_wh(depth, MIFX_FORMAT_F32, fx->w, fx->h, "depth", fx->cleanup_in.depth));
    MIFX_CHECK(to_img_wh(normal, MIFX_FORMAT_F32X4, fx->w, fx->h, "normal", fx->cleanup_normal));
    return fx->run_cleanup();
}

mifx_status mifx_ssr_get_intermediate(mifx_ssr* fx, const char* name, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && name != nullptr && out != nullptr, "mifx_ssr_get_intermediate: null argument");
    if (!fx->prepared || fx->last_frame == ~0u)
    {
        set_error("mifx_ssr_get_intermediate: nothing has been executed yet");
        return MIFX_ERR_INVALID_OP;
    }
    const int ci = int(fx->last_frame & 1u);
    const Plane* p = nullptr;
    int k = 0;
    if (std::sscanf(name, "hiz%d", &k) == 1 && k >= 1 && k < mifx_ssr::kMips) p = &fx->hiz[k];
    else if (!std::strcmp(name, "roughness")) p = &fx->roughness;
    else if (!std::strcmp(name, "mask")) p = &fx->mask;
    else if (!std::strcmp(name, "mask_half")) p = &fx->mask_half;
    else if (!std::strcmp(name, "ray_radiance")) p = &fx->ray_radiance;
    else if (!std::strcmp(name, "ray_dir_pdf")) p = &fx->ray_dir_pdf;
    else if (!std::strcmp(name, "res_radiance")) p = &fx->res_radiance;
    else if (!std::strcmp(name, "res_variance")) p = &fx->res_variance;
    else if (!std::strcmp(name, "res_depth")) p = &fx->res_depth;
    else if (!std::strcmp(name, "hist_radiance")) p = &fx->hist_radiance[ci];
    else if (!std::strcmp(name, "hist_variance")) p = &fx->hist_variance[ci];
    MIFX_REQUIRE(p != nullptr && p->data != nullptr, "mifx_ssr_get_intermediate: unknown or unallocated plane '%s'", name);
    *out = p->desc();
    return MIFX_OK;
}

} // extern "C"
