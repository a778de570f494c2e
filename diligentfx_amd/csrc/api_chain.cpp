// api_chain.cpp -- the canonical caller of the hot path: one frame of
//   PBR shade -> PostFX prep -> SSR -> SSAO -> composite -> TAA -> Bloom -> ToneMap
// in the order of HnPostProcessTask::Prepare / Execute (Hydrogent/src/Tasks/HnPostProcessTask.cpp:671-682, 743-948).
// SSR reads the un-composited scene colour of the current frame (FEATURE_FLAG_PREVIOUS_FRAME off), tone mapping is applied by the
// final copy-frame pass because TAA is on (HnPostProcessTask.cpp:172, :920-927).  Everything is recorded on the context stream.
#include "mifx_objects.h"
#include <cstdlib>

using namespace mifx;

mifx_chain::~mifx_chain()
{
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {evFork, evPrep, evSsao})
        if (e) (void)hipEventDestroy(e);
    if (side) (void)hipStreamDestroy(side);
    mifx_bloom_destroy(bloom);
    mifx_taa_destroy(taa);
    mifx_ssr_destroy(ssr);
    mifx_ssao_destroy(ssao);
    mifx_postfx_destroy(ctx);
}

extern "C" {

mifx_status mifx_chain_create(const mifx_device_desc* dev, const mifx_postfx_create_info* info, mifx_chain** out)
{
    MIFX_REQUIRE(dev != nullptr && out != nullptr, "mifx_chain_create: null argument");
    *out = nullptr;
    mifx_chain* c = new mifx_chain();
    mifx_status st = mifx_postfx_create(dev, info, &c->ctx);
    if (st >= 0) st = mifx_ssao_create(c->ctx, &c->ssao);
    if (st >= 0) st = mifx_ssr_create(c->ctx, &c->ssr);
    if (st >= 0) st = mifx_taa_create(c->ctx, &c->taa);
    if (st >= 0) st = mifx_bloom_create(c->ctx, &c->bloom);
    if (st < 0)
    {
        delete c;
        return st;
    }
    if (const char* e = std::getenv("MIFX_CHAIN_OVERLAP")) c->overlap = std::atoi(e) != 0;
    *out = c;
    return MIFX_OK;
}

void mifx_chain_destroy(mifx_chain* chain) { delete chain; }

mifx_status mifx_chain_get_postfx(mifx_chain* chain, mifx_postfx** out)
{
    MIFX_REQUIRE(chain != nullptr && out != nullptr, "mifx_chain_get_postfx: null argument");
    *out = chain->ctx;
    return MIFX_OK;
}

mifx_status mifx_chain_reset_history(mifx_chain* chain)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_reset_history: null argument");
    MIFX_CHECK(mifx_ssao_reset_history(chain->ssao));
    MIFX_CHECK(mifx_ssr_reset_history(chain->ssr));
    return mifx_taa_reset_history(chain->taa);
}

mifx_status mifx_chain_execute(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr)
{
    MIFX_REQUIRE(chain != nullptr && f != nullptr && out_ldr != nullptr, "mifx_chain_execute: null argument");
    MIFX_REQUIRE(f->curr_camera && f->prev_camera && f->ibl && f->pbr && f->ssao && f->ssr && f->taa && f->bloom && f->tone_mapping,
                 "mifx_chain_execute: every attribs pointer of mifx_chain_frame must be set");
    mifx_postfx* ctx = chain->ctx;
    const uint32_t W = f->frame.Width, H = f->frame.Height;
    // HnPostProcessTask::Prepare: per-frame PrepareResources in the order PostFX, SSAO, SSR, TAA, Bloom (:671-682)
    MIFX_CHECK(mifx_postfx_prepare(ctx, &f->frame, MIFX_POSTFX_FEATURE_FLAG_NONE));
    MIFX_CHECK(mifx_ssao_prepare(chain->ssao, ctx, MIFX_SSAO_FEATURE_FLAG_NONE));
    MIFX_CHECK(mifx_ssr_prepare(chain->ssr, ctx, MIFX_SSR_FEATURE_FLAG_NONE));
    MIFX_CHECK(mifx_taa_prepare(chain->taa, ctx, f->taa_feature_flags));
    MIFX_CHECK(mifx_bloom_prepare(chain->bloom, ctx, 0));
    MIFX_CHECK(chain->radiance.alloc(W, H, MIFX_FORMAT_F32X4));
    MIFX_CHECK(chain->specular_ibl.alloc(W, H, MIFX_FORMAT_F32X4));
    MIFX_CHECK(chain->composite.alloc(W, H, MIFX_FORMAT_F32X4));
    const mifx_image2d radiance = chain->radiance.desc(), spec = chain->specular_ibl.desc(), comp = chain->composite.desc();
    int stage = 0;
    auto mark = [&]() -> mifx_status {
        if (chain->profiling) MIFX_HIP_CHECK(hipEventRecord(chain->ev[stage], ctx->stream));
        ++stage;
        return MIFX_OK;
    };
    MIFX_CHECK(mark());

    mifx_postfx_render_attribs pa{f->gbuffer.depth, f->prev_depth, f->motion, f->curr_camera, f->prev_camera};
    mifx_ssr_render_attribs    sr{ctx, &radiance, f->gbuffer.depth, f->gbuffer.normal, f->gbuffer.material, f->motion, f->ssr};
    mifx_ssao_render_attribs   sa{ctx, f->gbuffer.depth, f->gbuffer.normal, f->ssao};
    if (chain->overlap && !chain->profiling)
    {
        // Two dependency chains:  shade -> SSR (needs the radiance and the prep outputs)   |   prep -> SSAO (depth / normals only).
        // The second one is recorded on the side stream; the launch stream joins before the composite.  Same kernels, same results.
        hipStream_t main = ctx->stream;
        if (!chain->side)
        {
            MIFX_HIP_CHECK(hipStreamCreateWithFlags(&chain->side, hipStreamNonBlocking));
            for (hipEvent_t* e : {&chain->evFork, &chain->evPrep, &chain->evSsao}) MIFX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
        }
        MIFX_HIP_CHECK(hipEventRecord(chain->evFork, main));
        MIFX_HIP_CHECK(hipStreamWaitEvent(chain->side, chain->evFork, 0));
        ctx->stream = chain->side;
        mifx_status st = mifx_postfx_execute(ctx, &pa);
        if (st >= 0) st = hipEventRecord(chain->evPrep, chain->side) == hipSuccess ? MIFX_OK : MIFX_ERR_HIP;
        if (st >= 0) st = mifx_ssao_execute(chain->ssao, &sa);
        if (st >= 0) st = hipEventRecord(chain->evSsao, chain->side) == hipSuccess ? MIFX_OK : MIFX_ERR_HIP;
        ctx->stream = main;
        MIFX_CHECK(st);
        MIFX_CHECK(mifx_pbr_shade_execute(ctx, &f->gbuffer, f->curr_camera, f->pbr, f->ibl, f->background, &radiance, &spec));
        MIFX_HIP_CHECK(hipStreamWaitEvent(main, chain->evPrep, 0));
        MIFX_CHECK(mifx_ssr_execute(chain->ssr, &sr));
        MIFX_HIP_CHECK(hipStreamWaitEvent(main, chain->evSsao, 0));
        stage += 4;
    }
    else
    {
        // forward shade (stands in for HnRenderRprimsTask: SceneColor + the IBL target of the USD G-buffer)
        MIFX_CHECK(mifx_pbr_shade_execute(ctx, &f->gbuffer, f->curr_camera, f->pbr, f->ibl, f->background, &radiance, &spec));
        MIFX_CHECK(mark());
        // PostFXContext::Execute (:788-809)
        MIFX_CHECK(mifx_postfx_execute(ctx, &pa));
        MIFX_CHECK(mark());
        // ScreenSpaceReflection::Execute (:811-822)
        MIFX_CHECK(mifx_ssr_execute(chain->ssr, &sr));
        MIFX_CHECK(mark());
        // ScreenSpaceAmbientOcclusion::Execute (:824-832)
        MIFX_CHECK(mifx_ssao_execute(chain->ssao, &sa));
        MIFX_CHECK(mark());
    }
    mifx_image2d ssr_out, ssao_out, taa_out, bloom_out;
    MIFX_CHECK(mifx_ssr_get_output(chain->ssr, &ssr_out));
    MIFX_CHECK(mifx_ssao_get_output(chain->ssao, &ssao_out));
    // composite draw (:834-869), no tone mapping while TAA is on
    mifx_composite_attribs ca{&radiance, &spec, &ssr_out, &ssao_out, f->gbuffer.normal, f->gbuffer.base_color, f->gbuffer.material, f->ibl->brdf_lut,
                              f->curr_camera, f->ssr_scale, f->ssao_scale, nullptr, f->ave_log_lum};
    MIFX_CHECK(mifx_composite_execute(ctx, &ca, &comp));
    MIFX_CHECK(mark());
    // TemporalAntiAliasing::Execute on the jittered composite (:871-897)
    mifx_taa_render_attribs ta{ctx, &comp, f->taa};
    MIFX_CHECK(mifx_taa_execute(chain->taa, &ta));
    MIFX_CHECK(mifx_taa_get_output(chain->taa, 0, &taa_out));
    MIFX_CHECK(mark());
    // Bloom::Execute on the TAA output (:911-918)
    mifx_bloom_render_attribs ba{ctx, &taa_out, f->bloom};
    MIFX_CHECK(mifx_bloom_execute(chain->bloom, &ba));
    MIFX_CHECK(mifx_bloom_get_output(chain->bloom, &bloom_out));
    MIFX_CHECK(mark());
    // copy-frame draw = ToneMap (+ sRGB) (:920-926)
    MIFX_CHECK(mifx_tonemap_execute(ctx, &bloom_out, out_ldr, f->tone_mapping, f->ave_log_lum, f->tonemap_flags));
    MIFX_CHECK(mark());
    chain->timed = chain->profiling;
    return MIFX_OK;
}

mifx_status mifx_chain_set_overlap(mifx_chain* chain, int32_t enable)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_overlap: null chain");
    chain->overlap = enable != 0;
    return MIFX_OK;
}

mifx_status mifx_chain_set_profiling(mifx_chain* chain, int32_t enable)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_profiling: null argument");
    MIFX_HIP_CHECK(hipSetDevice(chain->ctx->device));
    if (enable)
        for (auto& e : chain->ev)
            if (!e) MIFX_HIP_CHECK(hipEventCreate(&e));
    chain->profiling = enable != 0;
    chain->timed     = false;
    return MIFX_OK;
}

mifx_status mifx_chain_get_stage_times(mifx_chain* chain, float out_ms[MIFX_CHAIN_STAGE_COUNT])
{
    MIFX_REQUIRE(chain != nullptr && out_ms != nullptr, "mifx_chain_get_stage_times: null argument");
    if (!chain->timed)
    {
        set_error("mifx_chain_get_stage_times: enable profiling and execute a frame first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_HIP_CHECK(hipEventSynchronize(chain->ev[MIFX_CHAIN_STAGE_COUNT]));
    for (int i = 0; i < MIFX_CHAIN_STAGE_COUNT; ++i) MIFX_HIP_CHECK(hipEventElapsedTime(&out_ms[i], chain->ev[i], chain->ev[i + 1]));
    return MIFX_OK;
}

} // extern "C"
