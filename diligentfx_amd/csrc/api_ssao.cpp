// api_ssao.cpp -- C ABI + host sequencing of ScreenSpaceAmbientOcclusion
// (PostProcess/ScreenSpaceAmbientOcclusion/src/ScreenSpaceAmbientOcclusion.cpp: PrepareResources :61-346, Execute :348-387,
//  UpdateConstantBuffer :790-816, Compute* :818-1329).
//
// Differences from the reference that do not change results: mip 0 of the three pyramids are views of existing planes instead of
// copies (CopyTextureDepth :864, CopyTexture :1089-1106), the resolved-AO -> history copy (:1319-1328) is a second store of the resolve,
// background texels are written with the clear value by the kernels instead of clear + discard, and A7 + A8 run as one resolve over work lists
// (ssao.hip "fused resolve": the same value for every texel as the two full-frame passes, which remain behind mifx_debug_ssao_set_fused_resolve).
#include "mifx_objects.h"
#include <algorithm>
#include <cmath>
#include <cstdlib>

using namespace mifx;

extern "C" {

mifx_status mifx_ssao_create(mifx_postfx* ctx, mifx_ssao** out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_ssao_create: null argument");
    *out        = new mifx_ssao();
    (*out)->ctx = ctx;
    if (const char* e = std::getenv("MIFX_SSAO_FUSED_RESOLVE")) (*out)->fused_resolve = std::atoi(e) != 0;
    return MIFX_OK;
}

void mifx_ssao_destroy(mifx_ssao* fx) { delete fx; }

mifx_status mifx_ssao_prepare(mifx_ssao* fx, mifx_postfx* ctx, uint32_t feature_flags)
{
    MIFX_REQUIRE(fx != nullptr && ctx != nullptr, "mifx_ssao_prepare: null argument");
    if (!ctx->prepared)
    {
        set_error("mifx_ssao_prepare: mifx_postfx_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_REQUIRE((feature_flags & ~3u) == 0, "mifx_ssao_prepare: unknown feature flags 0x%x (ScreenSpaceAmbientOcclusion::FEATURE_FLAGS defines bits 0 and 1)", feature_flags);
    const bool half = (feature_flags & MIFX_SSAO_FEATURE_FLAG_HALF_RESOLUTION) != 0;
    MIFX_REQUIRE(!half || (ctx->frame.Width >= 32 && ctx->frame.Height >= 32), "mifx_ssao_prepare: frame too small for the half-resolution pyramid");
    fx->ctx = ctx;
    const uint32_t W = ctx->frame.Width, H = ctx->frame.Height;
    if (fx->prepared && fx->w == W && fx->h == H && fx->flags == feature_flags) return MIFX_OK;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    fx->prepared = false; // the object is ready again only when every plane of the new size exists: a failed allocation part-way must not leave
                          // `prepared` set with planes of the old size behind it (the next prepare / execute would then run out of bounds)
    // FEATURE_FLAG_HALF_RESOLUTION: the prefiltered depth pyramid and the AO target are (W / 2) x (H / 2) (.cpp:109-110, 273-274)
    const uint32_t AW = half ? W / 2u : W, AH = half ? H / 2u : H;
    for (int k = 1; k < mifx_ssao::kMips; ++k)
    {
        const uint32_t mw = (W >> k) ? (W >> k) : 1u, mh = (H >> k) ? (H >> k) : 1u;
        const uint32_t aw = (AW >> k) ? (AW >> k) : 1u, ah = (AH >> k) ? (AH >> k) : 1u;
        MIFX_CHECK(fx->prefiltered_depth[k].alloc(aw, ah, MIFX_FORMAT_F32));
        MIFX_CHECK(fx->conv_ao[k].alloc(mw, mh, MIFX_PLANE_AO));
        MIFX_CHECK(fx->conv_depth[k].alloc(mw, mh, MIFX_FORMAT_F32));
    }
    {
        size_t total = 0, off[mifx_ssao::kMips];
        uint32_t lw[mifx_ssao::kMips], lh[mifx_ssao::kMips], lp[mifx_ssao::kMips];
        for (int k = 0; k < mifx_ssao::kMips; ++k)
        {
            lw[k] = (AW >> k) ? (AW >> k) : 1u; lh[k] = (AH >> k) ? (AH >> k) : 1u;
            lp[k] = ((lw[k] * 4u + 255u) / 256u) * 256u;
            off[k] = total;
            total += size_t(lp[k]) * lh[k];
        }
        MIFX_REQUIRE(total < (size_t(1) << 32), "mifx_ssao_prepare: camera-z pyramid of %ux%u exceeds the 32-bit offset range", AW, AH);
        for (int k = 0; k < mifx_ssao::kMips; ++k) fx->prefiltered_camz[k].release();
        MIFX_CHECK(fx->camz_slab.reserve(total));
        for (int k = 0; k < mifx_ssao::kMips; ++k)
            fx->prefiltered_camz[k].attach(static_cast<unsigned char*>(fx->camz_slab.data) + off[k], lw[k], lh[k], lp[k], MIFX_FORMAT_F32);
    }
    // FEATURE_FLAG_HALF_PRECISION_DEPTH in the native-storage build: both depth pyramids are R16_UNORM targets in the reference (.cpp:95-97), their mip 0 a copy of the
    // depth into such a target (CopyTextureDepth, .cpp:857, 1131) -- a plane of its own here, with the values that copy keeps (mifx_device.h: depth16)
    fx->depth16 = mifx_storage_mode() == MIFX_STORAGE_RGBA16F && (feature_flags & MIFX_SSAO_FEATURE_FLAG_HALF_PRECISION_DEPTH) != 0;
    if (fx->depth16)
    {
        MIFX_CHECK(fx->prefiltered_depth[0].alloc(AW, AH, MIFX_FORMAT_F32));
        MIFX_CHECK(fx->conv_depth[0].alloc(W, H, MIFX_FORMAT_F32));
    }
    else
    {
        fx->prefiltered_depth[0].release();
        fx->conv_depth[0].release();
    }
    MIFX_CHECK(fx->occlusion.alloc(AW, AH, MIFX_PLANE_AO));
    if (half)
    {
        MIFX_CHECK(fx->checkerboard_depth.alloc(AW, AH, MIFX_FORMAT_F32));
        MIFX_CHECK(fx->full_camz.alloc(W, H, MIFX_FORMAT_F32));
        MIFX_CHECK(fx->occlusion_upsampled.alloc(W, H, MIFX_PLANE_AO));
    }
    else
    {
        fx->checkerboard_depth.release();
        fx->full_camz.release();
        fx->occlusion_upsampled.release();
    }
    MIFX_CHECK(fx->accum_ao.alloc(W, H, MIFX_PLANE_AO));
    MIFX_CHECK(fx->resampled.alloc(W, H, MIFX_PLANE_AO));
    if (fx->alias_output) fx->output.release();
    else
    {
        MIFX_CHECK(fx->output.alloc(W, H, MIFX_PLANE_AO));
        MIFX_CHECK(fx->output.fill(ctx->stream, 1.0f)); // cleared like the history targets it mirrors
    }
    MIFX_CHECK(fx->resolve_lists.reserve(ssao_resolve_list_bytes(W, H)));
    ctx->queued_outside_execute(); // (the fills below and above)
    for (int i = 0; i < 2; ++i)
    {
        MIFX_CHECK(fx->history_ao[i].alloc(W, H, MIFX_PLANE_AO));
        MIFX_CHECK(fx->history_len[i].alloc(W, H, MIFX_PLANE_HISTORY_LEN));
        // history targets are cleared to 1.0 when (re)created (.cpp:304-305, :320-321)
        MIFX_CHECK(fx->history_ao[i].fill(ctx->stream, 1.0f));
        MIFX_CHECK(fx->history_len[i].fill(ctx->stream, 1.0f));
    }
    fx->w = W; fx->h = H; fx->flags = feature_flags;
    // (last_frame is kept: the reference recreates and clears its targets on a resize or a flag change but keeps m_LastFrameIdx, so the next frame accumulates onto the
    //  cleared history -- AO 1, length 1 -- without ResetAccumulation (.cpp:65-96, 797-800); confirmed by executing the reference's host code, oracle/refhost)
    fx->prepared    = true;
    return MIFX_OK;
}

mifx_status mifx_ssao_reset_history(mifx_ssao* fx)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_ssao_reset_history: null argument");
    fx->last_frame  = ~0u;
    fx->force_reset = true;
    if (fx->ctx) fx->ctx->queued_outside_execute();
    if (fx->prepared)
        for (int i = 0; i < 2; ++i)
        {
            MIFX_CHECK(fx->history_ao[i].fill(fx->ctx->stream, 1.0f));
            MIFX_CHECK(fx->history_len[i].fill(fx->ctx->stream, 1.0f));
        }
    if (fx->prepared && fx->output.data) MIFX_CHECK(fx->output.fill(fx->ctx->stream, 1.0f));
    return MIFX_OK;
}

mifx_status mifx_debug_ssao_set_fused_resolve(mifx_ssao* fx, int32_t enable)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_debug_ssao_set_fused_resolve: null argument");
    fx->fused_resolve = enable != 0;
    return MIFX_OK;
}

mifx_status mifx_ssao_execute(mifx_ssao* fx, const mifx_ssao_render_attribs* ra)
{
    MIFX_REQUIRE(fx != nullptr && ra != nullptr && ra->attribs != nullptr, "mifx_ssao_execute: null argument");
    mifx_postfx* ctx = ra->postfx ? ra->postfx : fx->ctx;
    if (!fx->prepared || !ctx || !ctx->executed)
    {
        set_error("mifx_ssao_execute: call mifx_ssao_prepare and mifx_postfx_execute for this frame first");
        return MIFX_ERR_INVALID_OP;
    }
    if (!ctx->sobol_dev)
    {
        set_error("mifx_ssao_execute: the PostFX context has no blue-noise tables");
        return MIFX_ERR_INVALID_OP;
    }
    const uint32_t W = fx->w, H = fx->h;
    MIFX_REQUIRE(ctx->frame.Width == W && ctx->frame.Height == H, "mifx_ssao_execute: frame size changed without mifx_ssao_prepare");
    MIFX_REQUIRE(ra->attribs->Algorithm <= MIFX_SSAO_ALGORITHM_VBAO, "mifx_ssao_execute: unknown algorithm %u", ra->attribs->Algorithm);
    Img depth, normal;
    MIFX_CHECK(to_img_wh(ra->depth, MIFX_FORMAT_F32, W, H, "depth", depth));
    MIFX_CHECK(to_img_wh(ra->normal, MIFX_FORMAT_F32X4, W, H, "normal", normal));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;

    MIFX_RANGE("ScreenSpaceAmbientOcclusion");
    // UpdateConstantBuffer (.cpp:790-816): reset on the first frame, on a frame-index gap, or on request
    const uint32_t idx = ctx->frame.Index;
    mifx_ssao_attribs a = *ra->attribs;
    const bool reset = fx->force_reset || fx->last_frame == ~0u || idx != fx->last_frame + 1u || a.ResetAccumulation != 0;
    a.ResetAccumulation = reset ? 1 : 0;
    fx->last_frame  = idx;
    fx->force_reset = false;

    const bool rev = (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0; // SSAO_OPTION_INVERTED_DEPTH (ScreenSpaceAmbientOcclusion.cpp:72)
    const CamK cur = make_camk(ctx->curr_cam, rev), prev = make_camk(ctx->prev_cam, rev);
    const int  ci = int(idx & 1u), pi = int((idx + 1u) & 1u); // ping-pong (.cpp:1044-1045)
    Img prevDepth, dummy;
    MIFX_CHECK(to_img_wh(&ctx->prev_depth, MIFX_FORMAT_F32, W, H, "previous depth", prevDepth));
    (void)dummy;

    const bool half = (fx->flags & MIFX_SSAO_FEATURE_FLAG_HALF_RESOLUTION) != 0;
    // A1 (half resolution): checkerboard of the min / max depth of the 2x2 blocks (.cpp:818-838)
    if (half) MIFX_CHECK(launch_ssao_downsample_depth(s, depth, fx->checkerboard_depth.view()));
    // A2: prefiltered depth pyramid (mip 0 = the depth itself, or the checkerboard depth :857)
    Pyr dpyr{};
    dpyr.levels = mifx_ssao::kMips;
    dpyr.l[0]   = half ? fx->checkerboard_depth.view() : depth;
    if (fx->depth16)
    {
        MIFX_CHECK(launch_depth16_copy(s, dpyr.l[0], fx->prefiltered_depth[0].view()));
        dpyr.l[0] = fx->prefiltered_depth[0].view();
    }
    for (int k = 1; k < mifx_ssao::kMips; ++k) dpyr.l[k] = fx->prefiltered_depth[k].view();
    Pyr zpyr{};
    zpyr.levels = mifx_ssao::kMips;
    for (int k = 0; k < mifx_ssao::kMips; ++k) zpyr.l[k] = fx->prefiltered_camz[k].view();
    // Row-band sharding: the pyramids are built whole (a tap of A3 can land anywhere), except the camera z of level 0 -- 40 % of the pass's bytes -- which only
    // the taps that stay at level 0 read, i.e. those within sqrt(MipLenSq[0]) pixels of a pixel of A3's rows (tap_mip), and A8 on its own rows.
    // Round 5: the same holds level by level -- a tap at level k lies within 2^(k + 0.5 + offset) pixels of its pixel, only the last level is read anywhere -- so when one
    // launch reduces the whole pyramid (even sizes all the way: launch_ssao_prefilter_pyramid) the levels in between are STORED on those rows only; every level is still
    // computed whole on the way to the last one.  Nothing but A3 reads them.
    Pyr zbuild = zpyr, dbuild = dpyr;
    const Rows ownLast    = fx->own_last_level; // (per-frame requests: taken and cleared before anything can return)
    const bool gatherLast = fx->gather_last_level;
    auto       afterPrefilter = std::move(fx->after_prefilter);
    fx->own_last_level    = Rows{0, 0};
    fx->gather_last_level = false;
    fx->after_prefilter   = nullptr;
    if (!ctx->band.empty() && !half)
    {
        const double reach0 = std::exp2(0.5 + double(a.DepthMIPSamplingOffset)); // sqrt of the first threshold of tap_mip (mifx_effects.h: MipLenSq[0] = 2^(1 + 2 offset))
        const Rows   a3     = rows_expand(rows_align(rows_expand(rows_expand(ctx->needed_rows(int(H)), int(std::ceil(a.SpatialReconstructionRadius)) + 1, int(H)), 48, int(H)), mifx_ssao::kWindowAlign, int(H)), 1, int(H));
        zbuild.l[0] = win(zpyr.l[0], rows_align(rows_expand(a3, int(std::ceil(reach0)) + 2, int(H)), 2, int(H))); // (48: the wider of the two A7 reaches below -- a superset is always safe here)
        if (pyramid_fusable_levels(int(W), int(H), mifx_ssao::kMips - 1) == mifx_ssao::kMips - 1)
            for (int k = 1; k < mifx_ssao::kMips - 1; ++k)
            {
                const Rows r0 = rows_expand(a3, int(std::ceil(std::ldexp(reach0, k))) + 2, int(H)); // rows of the frame a level-k tap of A3's rows can fall on
                const Rows rk = rows_clip(Rows{(r0.b >> k) - 1, ((r0.e + (1 << k) - 1) >> k) + 1}, zpyr.l[k].h);
                zbuild.l[k] = win(zpyr.l[k], rk);
                dbuild.l[k] = win(dpyr.l[k], rk);
            }
        // Round 6: the last level comes from its owners (mifx_ssao::own_last_level / after_prefilter).  This rank then reduces the source rows that its store windows of
        // the levels in between and its own rows of the last level cover -- on a boundary of one row of the last level -- and stores the last level on those rows.
        if (gatherLast && pyramid_fusable_levels(int(W), int(H), mifx_ssao::kMips - 1) == mifx_ssao::kMips - 1 && !fx->depth16)
        {
            constexpr int kLast = mifx_ssao::kMips - 1;
            Rows need = Rows{zbuild.l[0].y0, row_end(zbuild.l[0])};
            if (!ownLast.empty()) need = Rows{std::min(need.b, ownLast.b << kLast), std::max(need.e, ownLast.e << kLast)};
            for (int k = 1; k < kLast; ++k)
            {
                const Rows rk{dbuild.l[k].y0 << k, row_end(dbuild.l[k]) << k};
                need = Rows{std::min(need.b, rk.b), std::max(need.e, rk.e)};
            }
            const Rows cw = rows_align(rows_clip(need, int(H)), 1 << kLast, int(H));
            dbuild.l[0]     = win(dpyr.l[0], cw);
            dbuild.l[kLast] = win(dpyr.l[kLast], Rows{cw.b >> kLast, (cw.e + (1 << kLast) - 1) >> kLast});
            zbuild.l[kLast] = win(zpyr.l[kLast], Rows{cw.b >> kLast, (cw.e + (1 << kLast) - 1) >> kLast});
        }
        else MIFX_REQUIRE(!gatherLast && !afterPrefilter, "mifx_ssao_execute: the last pyramid level cannot be gathered in this configuration (frame %ux%u)", W, H);
    }
    else MIFX_REQUIRE(!gatherLast && !afterPrefilter, "mifx_ssao_execute: gather_last_level without a row band / in half-resolution mode");
    MIFX_CHECK(launch_ssao_prefilter_pyramid(s, dbuild, zbuild, cur, a, fx->depth16));
    if (afterPrefilter) MIFX_CHECK(afterPrefilter(fx->prefiltered_depth[mifx_ssao::kMips - 1], fx->prefiltered_camz[mifx_ssao::kMips - 1], s));
    // Row windows (mifx_rows.h), from the rows of the output its consumers need back to the first pass; whole frame by default.
    //   A8 reads the resampled AO at Poisson taps of radius <= SpatialReconstructionRadius (|xi| <= 1, truncation: +1 row);
    //   A7 reads the box pyramids up to level 4 with 2x2 taps: a level-4 texel spans 16 rows and the two tap rows cover y - 23.5 .. y + 23.5 (24 rows) when the
    //   taps sit on texel centres (frame divisible by 16: ssao_resample_kernel<true>); the general linear tap reaches one texel further (32 + 16 rows);
    //   A6's last level has one row per 16 rows of the frame and the fused kernel reduces whole source blocks: the window is aligned to that (kWindowAlign);
    //   A5 reads the 3x3 neighbourhood of the current AO; its history taps are covered by the halo exchange of the history planes.
    const int  iH = int(H);
    const Rows w8 = ctx->needed_rows(iH);
    const Rows w7 = rows_expand(w8, int(std::ceil(a.SpatialReconstructionRadius)) + 1, iH);
    const bool centredTaps = int(cur.vw) == int(W) && int(cur.vh) == iH && W % 16u == 0u && H % 16u == 0u && !half; // the condition of launch_ssao_resample
    const Rows w5 = rows_align(rows_expand(w7, centredTaps ? 24 : 48, iH), mifx_ssao::kWindowAlign, iH);
    const Rows w3 = rows_expand(w5, 1, iH);
    MIFX_REQUIRE(ctx->prep_rows.empty() || rows_contain(ctx->prep_rows, w5), "mifx_ssao_execute: PostFX prep covered rows [%d, %d), needed [%d, %d)", ctx->prep_rows.b,
                 ctx->prep_rows.e, w5.b, w5.e);
    // A3
    {
        MifxKernelTimer timer(ctx, "ssao_compute_ao_kernel");
        // (half resolution: A4 reads the half-size AO at the rows int(y / 2) - 1 .. + 1 of every row y it writes -- w3, what A5's 3x3 statistic reads)
        const Rows h3 = rows_clip(Rows{w3.b / 2 - 1, (w3.e - 1) / 2 + 2}, int(fx->occlusion.h));
        MIFX_CHECK(launch_ssao_compute_ao(s, dpyr, zpyr, normal, ctx->noise_zw.view(), win(fx->occlusion.view(), half ? h3 : w3), cur, a, half,
                                          (fx->flags & MIFX_SSAO_FEATURE_FLAG_HALF_PRECISION_DEPTH) != 0));
    }
    // A4 (half resolution): bilateral upsampling guided by the full-size depth (.cpp:985-1008); A8 then needs the camera z of the full-size depth
    Img currAO = fx->occlusion.view(), fullCamz = zpyr.l[0];
    if (half)
    {
        MIFX_CHECK(launch_ssao_bilateral_upsample(s, depth, fx->occlusion.view(), win(fx->occlusion_upsampled.view(), w3), cur));
        MIFX_CHECK(launch_ssao_depth_to_camz(s, depth, win(fx->full_camz.view(), w7), cur)); // (read by A8's taps: the rows of the resampled AO)
        currAO   = fx->occlusion_upsampled.view();
        fullCamz = fx->full_camz.view();
    }
    // The resolved AO goes to history_ao[curr] (the reference copies it there, ScreenSpaceAmbientOcclusion.cpp:1319-1328) and, unless the chain aliased the two, to
    // the stable `output` plane.
    const Img hist = win(fx->history_ao[ci].view(), w8);
    const Img outp = fx->alias_output ? Img{} : win(fx->output.view(), w8);
    const Img acc5 = win(fx->accum_ao.view(), w5);
    SsaoResolve resolve{depth, win(fx->resampled.view(), w7), hist, outp, fx->resolve_lists.data};
    // (with R16_UNORM pyramids A7 tests the background on the convoluted pyramid's mip 0 and A8 on the depth buffer: two different values for a depth within half a code of
    //  the far plane, so the fused resolve, which shares one load between them, stays off)
    const bool fusedResolve = fx->fused_resolve && !fx->depth16;
    // A5 (:1047: the upsampled occlusion in half-resolution mode); with the fused resolve it also does A7's copy, A8's early path and fills the two work lists
    MifxKernelTimer t5(ctx, "ssao_temporal_kernel");
    MIFX_CHECK(launch_ssao_temporal(s, currAO, fx->history_ao[pi].view(), fx->history_len[pi].view(), ctx->reproj_depth.view(), prevDepth, ctx->closest_motion.view(), acc5,
                                    fx->history_len[ci].view(), cur, prev, a, fusedResolve ? &resolve : nullptr));
    t5.stop();
    // A6: box pyramids of the accumulated AO and of the depth (mip 0 = views)
    Pyr apyr{}, cdpyr{};
    apyr.levels = cdpyr.levels = mifx_ssao::kMips;
    apyr.l[0]  = fx->accum_ao.view();
    cdpyr.l[0] = depth;
    if (fx->depth16)
    {
        MIFX_CHECK(launch_depth16_copy(s, depth, fx->conv_depth[0].view()));
        cdpyr.l[0] = fx->conv_depth[0].view();
    }
    Rows wl = w5;
    for (int k = 1; k < mifx_ssao::kMips; ++k)
    {
        wl = Rows{wl.b / 2, (wl.e + 1) / 2}; // rows of level k computed from the rows wl of level k - 1 (w5 is aligned to kWindowAlign rows or clipped)
        apyr.l[k]  = win(fx->conv_ao[k].view(), wl);
        cdpyr.l[k] = win(fx->conv_depth[k].view(), wl);
    }
    MIFX_CHECK(launch_ssao_convolute_pyramids(s, apyr, cdpyr, fx->depth16));
    // A7 + A8
    if (fusedResolve)
    {
        MifxKernelTimer timer(ctx, "ssao_resolve_list_kernels");
        MIFX_CHECK(launch_ssao_resolve_lists(s, apyr, cdpyr, fx->history_len[ci].view(), fullCamz, normal, acc5, resolve, cur, a));
    }
    else
    {
        {
            MifxKernelTimer timer(ctx, "ssao_resample_kernel");
            MIFX_CHECK(launch_ssao_resample(s, apyr, cdpyr, fx->history_len[ci].view(), normal, win(fx->resampled.view(), w7), cur));
        }
        MifxKernelTimer t8(ctx, "ssao_spatial_kernel");
        MIFX_CHECK(launch_ssao_spatial(s, fx->resampled.view(), fx->history_len[ci].view(), depth, fullCamz, normal, fx->alias_output ? hist : outp, fx->alias_output ? Img{} : hist, cur, a));
        t8.stop();
    }
    return reset ? MIFX_NO_HISTORY : MIFX_OK;
}

mifx_status mifx_ssao_get_output(mifx_ssao* fx, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && out != nullptr, "mifx_ssao_get_output: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_ssao_get_output: resources are not prepared");
        return MIFX_ERR_INVALID_OP;
    }
    // GetAmbientOcclusionSRV: one plane for the life of the prepared object.  (Inside mifx_chain: the history plane of the last executed frame -- re-queried every frame.)
    *out = fx->alias_output ? fx->history_ao[fx->last_frame == ~0u ? 0u : (fx->last_frame & 1u)].desc() : fx->output.desc();
    return MIFX_OK;
}

mifx_status mifx_ssao_get_intermediate(mifx_ssao* fx, const char* name, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && name != nullptr && out != nullptr, "mifx_ssao_get_intermediate: null argument");
    if (!fx->prepared || fx->last_frame == ~0u)
    {
        set_error("mifx_ssao_get_intermediate: nothing has been executed yet");
        return MIFX_ERR_INVALID_OP;
    }
    const int ci = int(fx->last_frame & 1u);
    const Plane* p = nullptr;
    int k = 0;
    if (std::sscanf(name, "prefiltered_depth%d", &k) == 1 && k >= 1 && k < mifx_ssao::kMips) p = &fx->prefiltered_depth[k];
    else if (std::sscanf(name, "conv_ao%d", &k) == 1 && k >= 1 && k < mifx_ssao::kMips) p = &fx->conv_ao[k];
    else if (std::sscanf(name, "conv_depth%d", &k) == 1 && k >= 1 && k < mifx_ssao::kMips) p = &fx->conv_depth[k];
    else if (!std::strcmp(name, "occlusion")) p = &fx->occlusion;
    else if (!std::strcmp(name, "checkerboard_depth")) p = &fx->checkerboard_depth;
    else if (!std::strcmp(name, "occlusion_upsampled")) p = &fx->occlusion_upsampled;
    else if (!std::strcmp(name, "history_ao")) p = &fx->history_ao[ci];
    else if (!std::strcmp(name, "accum_ao")) p = &fx->accum_ao;
    else if (!std::strcmp(name, "history_len")) p = &fx->history_len[ci];
    else if (!std::strcmp(name, "resampled")) p = &fx->resampled;
    MIFX_REQUIRE(p != nullptr && p->data != nullptr, "mifx_ssao_get_intermediate: unknown or unallocated plane '%s'", name);
    *out = p->desc();
    return MIFX_OK;
}

} // extern "C"
