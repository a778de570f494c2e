// api_bloom_taa.cpp -- C ABI + host sequencing of Bloom (PostProcess/Bloom/src/Bloom.cpp) and TemporalAntiAliasing
// (PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp).
#include <cmath>

#include "mifx_objects.h"

using namespace mifx;

mifx_bloom::~mifx_bloom()
{
    for (auto* p : down) delete p;
    for (auto* p : up) delete p;
}

// DiligentCore ComputeMipLevelsCount: floor(log2(max(w, h))) + 1
static uint32_t compute_mip_levels_count(uint32_t w, uint32_t h)
{
    uint32_t m = w > h ? w : h, n = 0;
    while (m > 0) { ++n; m >>= 1; }
    return n;
}

extern "C" {

// ------------------------------------------------------------------------------------------------ Bloom
mifx_status mifx_bloom_create(mifx_postfx* ctx, mifx_bloom** out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_bloom_create: null argument");
    *out        = new mifx_bloom();
    (*out)->ctx = ctx;
    return MIFX_OK;
}
void mifx_bloom_destroy(mifx_bloom* fx) { delete fx; }

// Bloom::PrepareResources (Bloom.cpp:74-150): half-resolution pyramids of TextureCount = ComputeMipLevelsCount(W/2, H/2) levels
mifx_status mifx_bloom_prepare(mifx_bloom* fx, mifx_postfx* ctx, uint32_t feature_flags)
{
    MIFX_REQUIRE(fx != nullptr && ctx != nullptr, "mifx_bloom_prepare: null argument");
    if (!ctx->prepared)
    {
        set_error("mifx_bloom_prepare: mifx_postfx_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_REQUIRE(feature_flags == 0, "mifx_bloom_prepare: unknown feature flags 0x%x", feature_flags);
    fx->ctx = ctx;
    // Bloom.cpp:84-85: behind a temporal up-scaler the effect works at the output size
    const bool upscaled = (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_TEMPORAL_UPSCALING) != 0;
    const uint32_t W = upscaled ? ctx->frame.OutputWidth : ctx->frame.Width, H = upscaled ? ctx->frame.OutputHeight : ctx->frame.Height;
    MIFX_REQUIRE(W >= 8 && H >= 8, "mifx_bloom_prepare: frame %ux%u too small for the bloom pyramid", W, H);
    if (fx->prepared && fx->w == W && fx->h == H) return MIFX_OK;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    fx->prepared = false; // until every plane of the new size exists (a failed allocation must not leave the old size marked as ready)
    fx->output_deferred = false; // (a final up-sample left for later belonged to the old planes -- and to a colour plane of the old size)
    for (auto* p : fx->down) delete p;
    for (auto* p : fx->up) delete p;
    fx->down.clear();
    fx->up.clear();
    const uint32_t hw = W / 2u, hh = H / 2u;
    const uint32_t count = compute_mip_levels_count(hw, hh);
    for (uint32_t i = 0; i < count; ++i)
    {
        const uint32_t lw = (hw >> i) ? (hw >> i) : 1u, lh = (hh >> i) ? (hh >> i) : 1u;
        fx->down.push_back(new Plane());
        fx->up.push_back(new Plane());
        MIFX_CHECK(fx->down.back()->alloc(lw, lh, MIFX_PLANE_BLOOM));
        MIFX_CHECK(fx->up.back()->alloc(lw, lh, MIFX_PLANE_BLOOM));
    }
    MIFX_CHECK(fx->output.alloc(W, H, MIFX_PLANE_BLOOM)); // (native-storage build: R11G11B10_FLOAT like the levels, Bloom.cpp:137; the tone map and the auto exposure take it)
    fx->w = W; fx->h = H; fx->flags = feature_flags; fx->prepared = true;
    return MIFX_OK;
}

// Bloom::ComputeMipCount (Bloom.cpp:152-156)
int mifx_bloom::mip_count(const mifx_bloom_attribs& a) const { return int(a.Radius * float(compute_mip_levels_count(down[0]->w, down[0]->h))); }

// Row windows of the fine levels (mifx_rows.h).  Footprints: a 13-tap down-sample texel r reads source rows 2r - 2 .. 2r + 3 (taps at +-2
// texels around 2r + 1/2, bilinear: +1), an up-sample texel y reads coarse rows y/2 - 2 .. y/2 + 2 (3x3 tent, bilinear: +1); one extra row
// of slack per level.
mifx_bloom::Plan mifx_bloom::make_plan(Rows band, Rows need, int mipCount) const
{
    Plan p;
    const int H = int(h);
    if (band.empty() || (band.b <= 0 && band.e >= H) || mipCount - 1 <= kGatherLevel || mipCount > 16) return p; // whole frame (or too few / too many levels to split)
    p.G = kGatherLevel;
    const int G = p.G;
    p.up[0] = rows_coarser(need, 3, int(up[0]->h));
    for (int i = 1; i < G; ++i) p.up[i] = rows_coarser(p.up[i - 1], 3, int(up[i]->h));
    const int sh = G + 1; // level G row r covers the full-resolution rows [r << sh, (r + 1) << sh)
    p.own = rows_clip(Rows{(band.b + (1 << sh) - 1) >> sh, band.e >= H ? int(down[G]->h) : (band.e + (1 << sh) - 1) >> sh}, int(down[G]->h));
    p.down[G] = p.own;
    for (int i = G - 1; i >= 0; --i) p.down[i] = rows_hull(p.up[i], rows_finer(p.down[i + 1], 4, int(down[i]->h)));
    // level 0 row r covers the full-resolution rows [2 r, 2 r + 2): the rows whose first full-resolution row lies in the band
    p.own0     = rows_clip(Rows{(band.b + 1) >> 1, band.e >= H ? int(down[0]->h) : (band.e + 1) >> 1}, int(down[0]->h));
    p.compute0 = p.down[0];
    if (halo_level0 && G == 1 && int(down[0]->h) * 2 == H && !p.own0.empty()) p.compute0 = p.own0; // (the rows of down[0] beyond them come from their owners: after_level0)
    p.taa = rows_finer(p.compute0, 4, H);
    // Beyond the gathered level every rank holds down[G] whole, and computes of the coarser levels only the rows its band needs: the up-sample that writes
    // up[i - 1] reads down[i - 1] on the same rows and its coarser source (up[i]; down[last] for the last level) on rows_coarser(.., 3); down[i] also has to cover
    // what down[i + 1]'s window is reduced from.  (The windows grow by a few rows per level while the levels halve: from level ~5 on they are the whole level.)
    const int last = mipCount - 1;
    for (int i = G > 0 ? G : 1; i <= last; ++i) p.up[i] = rows_coarser(p.up[i - 1], 3, int(up[i]->h)); // (up[last] = the rows of down[last] the last up-sample reads)
    if (G == 0) p.up[0] = rows_coarser(need, 3, int(up[0]->h));
    p.down[last] = p.up[last];
    for (int i = last - 1; i > G; --i) p.down[i] = rows_hull(p.up[i], rows_finer(p.down[i + 1], 4, int(down[i]->h)));
    return p;
}

// Bloom::Execute (Bloom.cpp:407-446): prefilter (:288-311), downsample loop (:313-337), upsample loop + final composite (:339-396)
mifx_status mifx_bloom::run(const mifx_bloom_render_attribs* ra, int phase, const FusedToneMap* tone_map)
{
    MIFX_RANGE("Bloom");
    mifx_postfx* c = ra->postfx ? ra->postfx : ctx;
    // the source: the TAA output -- or, native-storage build, depth of field's output, an R11G11B10_FLOAT plane like Bloom's own (DepthOfField.cpp:281-289)
    Img  color;
    bool packed = false;
    MIFX_CHECK(to_img_hdr(ra->color, "color", color, packed));
    MIFX_REQUIRE(uint32_t(color.w) == w && uint32_t(color.h) == h, "mifx_bloom_execute: color is %dx%d, the effect was prepared for %ux%u", color.w, color.h, w, h);
    const mifx_bloom_attribs& a = *ra->attribs;
    const int mipCount = mip_count(a);
    MIFX_REQUIRE(mipCount >= 2 && mipCount <= int(down.size()),
                 "mifx_bloom_execute: Radius %g gives %d pyramid levels; the reference reads an unwritten texture below 2 levels", a.Radius, mipCount);
    MIFX_HIP_CHECK(hipSetDevice(c->device));
    hipStream_t s = c->stream;
    const Rows need = c->needed_rows(int(h));
    const Plan p    = make_plan(c->band, need, mipCount);
    MIFX_REQUIRE(phase == 0 || p.G >= 0, "mifx_bloom_execute: phased execution needs a row band (mifx_chain_set_row_band)");
    MIFX_REQUIRE(phase != 0 || p.G < 0, "mifx_bloom_execute: a row band is set; run the two phases around the gather of down[%d]", p.G);
    auto dwin = [&](int i) { return p.G >= 0 ? win(down[i]->view(), p.down[i]) : down[i]->view(); };
    auto uwin = [&](int i) { return p.G >= 0 ? win(up[i]->view(), p.up[i]) : up[i]->view(); };
    const int last = mipCount - 1;
    // The small levels (<= kTailTexels texels) go down and up again in ONE workgroup (launch_bloom_tail) instead of two dispatches each; with a row band only
    // the levels beyond the gathered one, which every rank computes whole.  `first` = the first level of that tail; no tail when it would hold a single level.
    int first = mipCount;
    for (int i = 1; i < mipCount; ++i)
        if (down[i]->w * down[i]->h <= kTailTexels) { first = i; break; }
    // (with a row band every level is computed on a row window, by the per-level kernels: the tail kernel takes whole levels)
    bool tail = p.G < 0 && fuse_tail && last >= first + 1 && last - first + 2 <= 8;
    Img  tailDown[8], tailUp[8];
    if (tail)
    {
        for (int i = first - 1; i <= last; ++i) { tailDown[i - first + 1] = down[i]->view(); tailUp[i - first + 1] = up[i]->view(); }
        tail = bloom_tail_fits(tailDown, last - first + 2); // (extreme aspect ratios halve one dimension only: the levels then shrink too slowly for the LDS budget)
    }
    const int  wide = tail ? first : mipCount; // levels below `wide` are produced by the per-level kernels
    if (phase != 2)
    {
        {
            MifxKernelTimer timer(c, "bloom_prefilter_kernel");
            MIFX_CHECK(launch_bloom_prefilter(s, color, p.G >= 0 ? win(down[0]->view(), p.compute0) : down[0]->view(), a, packed));
        }
        if (p.G >= 0 && after_level0) // (the rows of level 0 this rank reads but does not own: from the ranks that do)
        {
            auto hook = std::move(after_level0);
            after_level0 = nullptr;
            MIFX_CHECK(hook(*down[0], s));
        }
        for (int i = 1; i < wide && (p.G < 0 || i <= p.G); ++i) MIFX_CHECK(launch_bloom_downsample(s, down[i - 1]->view(), dwin(i)));
        if (phase == 1) return MIFX_OK; // the caller now assembles down[G] from all ranks
    }
    if (p.G >= 0)
        for (int i = p.G + 1; i < wide; ++i) MIFX_CHECK(launch_bloom_downsample(s, down[i - 1]->view(), dwin(i)));
    if (tail) MIFX_CHECK(launch_bloom_tail(s, tailDown, tailUp, last - first + 2));
    for (int i = tail ? first : last; i > 0; --i)
        MIFX_CHECK(launch_bloom_upsample(s, down[i - 1]->view(), i != last ? up[i]->view() : down[i]->view(), uwin(i - 1), a, false));
    if (tone_map)
    {
        MIFX_REQUIRE(tone_map->attribs->iToneMappingMode >= 0 && tone_map->attribs->iToneMappingMode <= MIFX_TONE_MAPPING_MODE_COMMERCE, "unknown tone mapping mode %d",
                     tone_map->attribs->iToneMappingMode);
        MIFX_REQUIRE((tone_map->flags & ~uint32_t(MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB)) == 0, "unknown tone map flags 0x%x", tone_map->flags);
        Img ldr;
        MIFX_CHECK(to_img_wh(tone_map->ldr, MIFX_FORMAT_F32X4, w, h, "ldr_out", ldr));
        MifxKernelTimer timer(c, "bloom_upsample_tonemap_kernel");
        MIFX_CHECK(launch_bloom_final_tonemap(s, color, up[0]->view(), win(output.view(), need), ldr, a, *tone_map->attribs, tone_map->ave_log_lum, tone_map->flags, !tone_map->skip_output, packed));
        output_deferred  = tone_map->skip_output;
        deferred_color   = color;
        deferred_packed  = packed;
        deferred_attribs = a;
        deferred_rows    = need;
    }
    else
    {
        MifxKernelTimer timer(c, "bloom_upsample_kernel");
        MIFX_CHECK(launch_bloom_upsample(s, color, up[0]->view(), win(output.view(), need), a, true, packed));
        output_deferred = false;
    }
    return MIFX_OK;
}

// the Bloom output of the last frame, when its final pass wrote the tone-mapped frame only: the plain final up-sample on the same inputs (the same texel arithmetic)
mifx_status mifx_bloom::run_deferred_output()
{
    if (!output_deferred) return MIFX_OK;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    // (queued on the context's stream outside an execute: a chain in a multi-stream mode orders the lanes of its next frame -- whose TAA / depth of field overwrite the
    //  colour plane this pass reads -- behind it: mifx_postfx::stream_epoch)
    ctx->queued_outside_execute();
    MIFX_CHECK(launch_bloom_upsample(ctx->stream, deferred_color, up[0]->view(), win(output.view(), deferred_rows), deferred_attribs, true, deferred_packed));
    output_deferred = false;
    return MIFX_OK;
}

mifx_status mifx_bloom_execute(mifx_bloom* fx, const mifx_bloom_render_attribs* ra)
{
    MIFX_REQUIRE(fx != nullptr && ra != nullptr && ra->attribs != nullptr, "mifx_bloom_execute: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_bloom_execute: mifx_bloom_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    return fx->run(ra, 0);
}

mifx_status mifx_bloom_get_output(mifx_bloom* fx, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && out != nullptr, "mifx_bloom_get_output: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_bloom_get_output: resources are not prepared");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_CHECK(fx->run_deferred_output()); // (a chain with MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND did not write the plane)
    *out = fx->output.desc();
    return MIFX_OK;
}

mifx_status mifx_debug_bloom_set_tail(mifx_bloom* fx, int32_t enable)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_debug_bloom_set_tail: null argument");
    fx->fuse_tail = enable != 0;
    return MIFX_OK;
}

mifx_status mifx_bloom_get_intermediate(mifx_bloom* fx, const char* name, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && name != nullptr && out != nullptr, "mifx_bloom_get_intermediate: null argument");
    int k = -1;
    const Plane* p = nullptr;
    if (fx->prepared && std::sscanf(name, "down%d", &k) == 1 && k >= 0 && k < int(fx->down.size())) p = fx->down[k];
    else if (fx->prepared && std::sscanf(name, "up%d", &k) == 1 && k >= 0 && k < int(fx->up.size())) p = fx->up[k];
    MIFX_REQUIRE(p != nullptr, "mifx_bloom_get_intermediate: unknown plane '%s'", name);
    *out = p->desc();
    return MIFX_OK;
}

// ------------------------------------------------------------------------------------------------ TemporalAntiAliasing
mifx_status mifx_taa_create(mifx_postfx* ctx, mifx_taa** out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_taa_create: null argument");
    *out        = new mifx_taa();
    (*out)->ctx = ctx;
    return MIFX_OK;
}
void mifx_taa_destroy(mifx_taa* fx) { delete fx; }

// TemporalAntiAliasing::PrepareResources / AccumulationBufferInfo::Prepare (TemporalAntiAliasing.cpp:80-121,145-167): two RGBA accumulation
// buffers, cleared to 0 on (re)creation (:112-118)
mifx_status mifx_taa_prepare(mifx_taa* fx, mifx_postfx* ctx, uint32_t feature_flags)
{
    MIFX_REQUIRE(fx != nullptr && ctx != nullptr, "mifx_taa_prepare: null argument");
    if (!ctx->prepared)
    {
        set_error("mifx_taa_prepare: mifx_postfx_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_REQUIRE((feature_flags & ~7u) == 0, "mifx_taa_prepare: unknown feature flags 0x%x", feature_flags);
    fx->ctx = ctx;
    const uint32_t W = ctx->frame.Width, H = ctx->frame.Height;
    fx->curr_frame = ctx->frame.Index;
    // AccumulationBufferInfo::Prepare recreates and clears the buffers on a size change only (TemporalAntiAliasing.cpp:84-88): switching BICUBIC / YCOCG /
    // GAUSSIAN at run time keeps the history, exactly as in the reference
    if (!(fx->prepared && fx->w == W && fx->h == H))
    {
        MIFX_HIP_CHECK(hipSetDevice(ctx->device));
        fx->prepared = false;
        ctx->queued_outside_execute();
        for (int i = 0; i < 2; ++i)
        {
            MIFX_CHECK(fx->accum[i].alloc(W, H, MIFX_FORMAT_F32X4));
            MIFX_CHECK(fx->accum[i].fill(ctx->stream, 0.0f));
        }
        fx->w = W; fx->h = H;
        // (no reset of last_frame: the reference keeps LastFrameIdx across a resize -- the next frame accumulates onto the cleared buffers, alpha 0 = no history weight)
        fx->prepared   = true;
    }
    fx->flags = feature_flags;
    fx->technique_ready = ((fx->techniques_created >> feature_flags) & 1u) != 0; // m_AllPSOsReady (:161-171)
    return MIFX_OK;
}

mifx_status mifx_taa_reset_history(mifx_taa* fx)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_taa_reset_history: null argument");
    // as far as results go, the state of a newly created object: no previous frame, and no flag set executed yet -- the next frame is again the placeholder frame of
    // its flag set (mifx_objects.h techniques_created), so that a sequence replayed after a reset reproduces the first run
    fx->last_frame         = ~0u;
    fx->techniques_created = 0;
    fx->technique_ready    = false;
    return MIFX_OK;
}

// TemporalAntiAliasing::Execute (TemporalAntiAliasing.cpp:169-201, ComputeTemporalAccumulation :260-300)
mifx_status mifx_taa_execute(mifx_taa* fx, const mifx_taa_render_attribs* ra)
{
    MIFX_REQUIRE(fx != nullptr && ra != nullptr && ra->attribs != nullptr, "mifx_taa_execute: null argument");
    const mifx::TaaFusedComposite* fused = fx->fused_composite; // (a per-frame request: taken and cleared before anything can return)
    fx->fused_composite = nullptr;
    mifx_postfx* ctx = ra->postfx ? ra->postfx : fx->ctx;
    if (!fx->prepared || !ctx || !ctx->executed)
    {
        // "TemporalAntiAliasing::PrepareResources must be called ..." LOG_ERROR + return in the reference (:178-184)
        set_error("mifx_taa_execute: call mifx_taa_prepare and mifx_postfx_execute for this frame first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_RANGE("TemporalAccumulation");
    const uint32_t W = fx->w, H = fx->h;
    Img color, prevDepth;
    MIFX_CHECK(to_img_wh(ra->color, MIFX_FORMAT_F32X4, W, H, "color", color));
    MIFX_CHECK(to_img_wh(&ctx->prev_depth, MIFX_FORMAT_F32, W, H, "previous depth", prevDepth));
    // AccumulationBufferInfo::UpdateConstantBuffer (:123-143)
    mifx_taa_attribs a = *ra->attribs;
    const uint32_t idx = fx->curr_frame;
    const bool reset = fx->last_frame == ~0u || idx != fx->last_frame + 1u || a.ResetAccumulation != 0;
    a.ResetAccumulation = reset ? 1 : 0;
    fx->last_frame = idx;
    const int ci = int(idx & 1u), pi = int((idx + 1u) & 1u); // :272-274, :293
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    fx->techniques_created |= 1u << fx->flags; // PrepareShadersAndPSO (:184)
    if (!fx->technique_ready)
    {
        MIFX_REQUIRE(fused == nullptr, "mifx_taa_execute: the placeholder frame copies the colour plane, which a fused composite does not write");
        // ComputePlaceholderTexture (:302-311): PostFXContext::CopyTextureColor of the colour buffer into the accumulation buffer of this frame
        const Rows   rows = ctx->needed_rows(int(H));
        const size_t row  = size_t(W) * texel_size(storage_format(MIFX_FORMAT_F32X4));
        const Plane& dst  = fx->accum[ci];
        if (!rows.empty())
            MIFX_HIP_CHECK(hipMemcpy2DAsync(static_cast<unsigned char*>(dst.data) + size_t(rows.b) * dst.pitch, dst.pitch, color.p + size_t(rows.b) * size_t(color.pitch), size_t(color.pitch), row,
                                            size_t(rows.e - rows.b), hipMemcpyDeviceToDevice, ctx->stream));
        return MIFX_NO_HISTORY;
    }
    {
        MifxKernelTimer timer(ctx, "taa_kernel");
        MIFX_CHECK(launch_taa(ctx->stream, color, fx->accum[pi].view(), ctx->closest_motion.view(), ctx->reproj_depth.view(), prevDepth,
                              win(fx->accum[ci].view(), ctx->needed_rows(int(fx->h))),
                              make_camk(ctx->curr_cam), make_camk(ctx->prev_cam), a, fx->flags, fused));
    }
    return reset ? MIFX_NO_HISTORY : MIFX_OK;
}

// GetAccumulatedFrameSRV (:203-214): BuffIdx = (CurrentFrameIdx + (IsPrevFrame ? 1 : 0)) & 1
mifx_status mifx_taa_get_output(mifx_taa* fx, int32_t is_prev_frame, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && out != nullptr, "mifx_taa_get_output: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_taa_get_output: resources are not prepared");
        return MIFX_ERR_INVALID_OP;
    }
    *out = fx->accum[(fx->curr_frame + (is_prev_frame ? 1u : 0u)) & 1u].desc();
    return MIFX_OK;
}

// HaltonSequence (:43-56) + GetJitterOffset (:63-78)
static float halton(uint32_t base, uint32_t index)
{
    float result = 0.0f, f = 1.0f;
    while (index > 0)
    {
        f      = f / float(base);
        result = result + f * float(index % base);
        index  = uint32_t(floorf(float(index) / float(base)));
    }
    return result;
}
mifx_status mifx_taa_get_jitter_offset(uint32_t frame_index, uint32_t width, uint32_t height, float out_jitter[2])
{
    MIFX_REQUIRE(out_jitter != nullptr, "mifx_taa_get_jitter_offset: null argument");
    if (width == 0 || height == 0)
    {
        out_jitter[0] = out_jitter[1] = 0.0f;
        return MIFX_OK;
    }
    const uint32_t kSamples = 16u;
    out_jitter[0] = (halton(2u, (frame_index % kSamples) + 1u) - 0.5f) / (0.5f * float(width));
    out_jitter[1] = (halton(3u, (frame_index % kSamples) + 1u) - 0.5f) / (0.5f * float(height));
    return MIFX_OK;
}
// GetJitteredProjMatrix (TemporalAntiAliasing.hpp:138-155)
mifx_status mifx_taa_get_jittered_proj_matrix(const float proj[16], const float jitter[2], float out_proj[16])
{
    MIFX_REQUIRE(proj != nullptr && jitter != nullptr && out_proj != nullptr, "mifx_taa_get_jittered_proj_matrix: null argument");
    for (int i = 0; i < 16; ++i) out_proj[i] = proj[i];
    if (proj[15] == 0.0f) { out_proj[8] += jitter[0]; out_proj[9] += jitter[1]; }   // perspective: m20 / m21
    else                  { out_proj[12] += jitter[0]; out_proj[13] += jitter[1]; } // orthographic: m30 / m31
    return MIFX_OK;
}

} // extern "C"
