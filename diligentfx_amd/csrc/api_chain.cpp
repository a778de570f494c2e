// api_chain.cpp -- the canonical caller of the hot path: one frame of
//   PBR shade -> PostFX prep -> SSR -> SSAO -> composite -> TAA -> Bloom -> ToneMap
// in the order of HnPostProcessTask::Prepare / Execute (Hydrogent/src/Tasks/HnPostProcessTask.cpp:671-682, 743-948).
// SSR reads the un-composited scene colour of the current frame (FEATURE_FLAG_PREVIOUS_FRAME off), tone mapping is applied by the
// final copy-frame pass because TAA is on (HnPostProcessTask.cpp:172, :920-927).  Everything is recorded on the context stream.
#include "mifx_objects.h"
#include <cstdlib>
#include <cmath>
#include <string>

using namespace mifx;

mifx_chain::~mifx_chain()
{
    for (auto& e : ev)
        if (e) (void)hipEventDestroy(e);
    if (halo_stream) (void)hipStreamSynchronize(halo_stream);
    if (ctx) ctx->pending_joins.clear();
    for (hipEvent_t e : {evFork, evPrep, evSsao, evPrepConsumed, evBloomDone, evJoinS, evJoinX, evAfterP1, evAfterP2, evHaloSsao, evHaloRest, evXEnd[0], evXEnd[1], evHiz, evJoinH, evSsrDone})
        if (e) (void)hipEventDestroy(e);
    for (auto& kv : signals)
        for (hipEvent_t e : kv.second.ev)
            if (e) (void)hipEventDestroy(e);
    if (ctx) ctx->kernel_hook = nullptr;
    if (halo_stream) (void)hipStreamDestroy(halo_stream);
    if (side) (void)hipStreamDestroy(side);
    if (lane_x) (void)hipStreamDestroy(lane_x);
    if (lane_h) (void)hipStreamDestroy(lane_h);
    mifx::chain_detach_comm(this);
    mifx_autoexposure_destroy(auto_exposure);
    mifx_bloom_destroy(bloom);
    mifx_dof_destroy(dof);
    mifx_taa_destroy(taa);
    mifx_ssr_destroy(ssr);
    mifx_ssao_destroy(ssao);
    mifx_postfx_destroy(ctx);
}

void mifx_chain::join_halos()
{
    // Only the context's stream is ordered behind the exchange here.  A lanes frame that follows (sharded or not, overlap >= 2) runs its first phases on the side
    // streams and, while chain_lanes_continue() holds, starts them behind the previous frame's events alone -- with the pending flags cleared nothing would order those
    // streams behind the ghost rows still arriving on halo_stream.  So the next frame forks every lane from the context's stream again.
    if (halo_ssao_pending || halo_rest_pending) prep_consumed = false;
    if (halo_ssao_pending) (void)hipStreamWaitEvent(ctx->stream, evHaloSsao, 0);
    if (halo_rest_pending) (void)hipStreamWaitEvent(ctx->stream, evHaloRest, 0);
    halo_ssao_pending = halo_rest_pending = false;
    ctx->pending_joins.clear();
}

extern "C" {

mifx_status mifx_chain_create(const mifx_device_desc* dev, const mifx_postfx_create_info* info, mifx_chain** out)
{
    MIFX_REQUIRE(dev != nullptr && out != nullptr, "mifx_chain_create: null argument");
    *out = nullptr;
    mifx_chain* c = new mifx_chain();
    mifx_status st = mifx_postfx_create(dev, info, &c->ctx);
    if (st >= 0) st = mifx_ssao_create(c->ctx, &c->ssao);
    if (st >= 0) c->ssao->alias_output = true; // the chain reads the AO of the frame it just executed: history_ao[curr] is the output (mifx_objects.h)
    if (st >= 0) st = mifx_ssr_create(c->ctx, &c->ssr);
    if (st >= 0) st = mifx_taa_create(c->ctx, &c->taa);
    if (st >= 0) st = mifx_bloom_create(c->ctx, &c->bloom);
    if (st < 0)
    {
        delete c;
        return st;
    }
    if (const char* e = std::getenv("MIFX_CHAIN_OVERLAP")) c->overlap = std::atoi(e) < 0 ? 0 : std::atoi(e) > 5 ? 5 : std::atoi(e);
    if (const char* e = std::getenv("MIFX_LANE_EDGES")) (void)mifx_chain_set_lane_edges(c, e); // (a malformed list is reported by the call itself when made directly)
    if (const char* e = std::getenv("MIFX_SHARD_ASYNC_HALOS")) c->async_halos = std::atoi(e) != 0;
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

mifx_status mifx_chain_get_effect(mifx_chain* chain, const char* name, void** out)
{
    MIFX_REQUIRE(chain != nullptr && name != nullptr && out != nullptr, "mifx_chain_get_effect: null argument");
    const std::string n = name;
    if (n == "ssao") *out = chain->ssao;
    else if (n == "ssr") *out = chain->ssr;
    else if (n == "taa") *out = chain->taa;
    else if (n == "bloom") *out = chain->bloom;
    else if (n == "dof") *out = chain->dof; // NULL until mifx_chain_set_depth_of_field enabled it
    else
    {
        set_error("mifx_chain_get_effect: unknown effect '%s'", name);
        return MIFX_ERR_INVALID_ARG;
    }
    return MIFX_OK;
}

mifx_status mifx_chain_reset_history(mifx_chain* chain)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_reset_history: null argument");
    chain->prep_consumed = false; // (the fills below are queued on the context stream: the next frame's lanes fork from it again)
    chain->join_halos();
    MIFX_CHECK(mifx_ssao_reset_history(chain->ssao));
    MIFX_CHECK(mifx_ssr_reset_history(chain->ssr));
    if (chain->dof) MIFX_CHECK(mifx_dof_reset_history(chain->dof)); // (the temporal circle of confusion: cleared as at creation -- the reference has no reset for it)
    if (chain->auto_exposure) MIFX_CHECK(mifx_autoexposure_reset(chain->auto_exposure, 0.1f)); // (the adapted average luminance back at the reference's start value)
    return mifx_taa_reset_history(chain->taa);
}

// The forward shade of the unsharded frame.  With fuse_ssr_mask the kernel also writes SSR's roughness / reflection-mask planes (pass R2 reads the same material
// and depth texels and nothing else): mifx_ssr_execute then finds them done for this frame.
// With a row band the shade runs on the rows of the composite; the by-product is needed on the rows of the ray march, a few rows more: the shade then covers
// those (+3 % of its rows against a pass over the material and depth planes).
static mifx_status chain_shade(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* radiance, const mifx_image2d* spec)
{
    mifx_postfx* ctx = chain->ctx;
    mifx_ssr*    ssr = chain->ssr;
    if (chain->has_layers || chain->has_shadows) // mifx_chain_set_material_layers: the layered kernel (no R2 by-product: SSR runs the pass itself)
    {
        const mifx_pbr_layers none{};
        chain->shaded_rows  = ctx->needed_rows(int(radiance->height));
        chain->shaded_frame = f->frame.Index;
        return mifx_pbr_shade_execute_layers(ctx, &f->gbuffer, chain->has_layers ? &chain->layers : &none, f->curr_camera, f->pbr, f->ibl, chain->has_shadows ? &chain->shadows : nullptr,
                                             f->background, radiance, spec);
    }
    if (!chain->fuse_ssr_mask || f->ssr->RoughnessChannel > 3u)
    {
        chain->shaded_rows  = ctx->needed_rows(int(radiance->height));
        chain->shaded_frame = f->frame.Index;
        return mifx_pbr_shade_execute(ctx, &f->gbuffer, f->curr_camera, f->pbr, f->ibl, f->background, radiance, spec);
    }
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    const int  H    = int(radiance->height);
    const Rows rows = ctx->band.empty() ? ctx->needed_rows(H) : mifx_ssr::march_rows(*f->ssr, ctx->needed_rows(H), H, (chain->ssr_flags & MIFX_SSR_FEATURE_FLAG_HALF_RESOLUTION) != 0);
    chain->shaded_rows  = rows;
    chain->shaded_frame = f->frame.Index;
    SsrMaskOut r2{ssr->roughness.view(), ssr->mask.view(), f->ssr->RoughnessThreshold, f->ssr->IsRoughnessPerceptual, f->ssr->RoughnessChannel, 1};
    MifxKernelTimer timer(ctx, "pbr_shade_ssr_mask_kernel"); // (includes the two cube-apron launches of the call)
    MIFX_CHECK(launch_pbr_shade(ctx->stream, ctx->ibl_apron, &f->gbuffer, *f->curr_camera, *f->pbr, f->ibl, f->background, radiance, spec, rows.b, rows.e,
                                (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0, nullptr, &r2));
    ssr->mask_provided_for = f->frame.Index;
    return MIFX_OK;
}

// The chain has ONE material plane, which the shade, SSR (pass R2) and the composite all read as the USD G-buffer's Material target (PerceptualRoughness, Metallic:
// USD_Renderer.cpp:98).  A specular-glossiness shade would read the same plane as PhysicalDesc (specular colour + glossiness) and SSR / the composite would then take
// their roughness from the specular-colour channels: refused.  A caller with specular-glossiness inputs shades with mifx_pbr_shade_execute and hands the chain's
// other effects the plane mifx_pbr_specgloss_to_material produces.
static mifx_status chain_check_workflow(const mifx_chain_frame* f)
{
    MIFX_REQUIRE(f->pbr->Workflow == MIFX_PBR_WORKFLOW_METALLIC_ROUGHNESS,
                 "mifx_chain_execute: Workflow %d: the chain's material plane is the metallic-roughness Material target (see mifx_pbr_specgloss_to_material)", f->pbr->Workflow);
    return MIFX_OK;
}

// The composite draw (HnPostProcess.psh:145-185).  With fuse_ssr_cleanup the kernel evaluates SSR's last pass (R7, the bilateral cleanup) for its own pixel from the
// effect's accumulated radiance instead of reading the plane R7 would have written (mifx_ssr_execute stopped after R6: mifx_objects.h `defer_cleanup`).
// With fuse_composite_taa on top of that (round 5) nothing is launched here: the TAA kernel of this frame evaluates the composite for the texels of its colour tile
// (taa.hip) -- the request is left in chain->pending_fused and handed to mifx_taa_execute by chain_taa below.  Not on TAA's placeholder frame (a copy of the plane).
static mifx_status chain_composite(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* radiance, const mifx_image2d* spec, const mifx_image2d* ssao_out,
                                   const mifx_image2d* comp)
{
    mifx_postfx* ctx = chain->ctx;
    mifx_ssr*    ssr = chain->ssr;
    mifx_image2d ssr_out{};
    const bool fused = ssr->cleanup_pending; // (set by an execute that deferred the pass)
    chain->pending_fused = mifx::TaaFusedComposite{nullptr, nullptr};
    if (!fused) MIFX_CHECK(mifx_ssr_get_output(ssr, &ssr_out));
    mifx_composite_attribs ca{radiance, spec, fused ? radiance /* not read */ : &ssr_out, ssao_out, f->gbuffer.normal, f->gbuffer.base_color, f->gbuffer.material, f->ibl->brdf_lut,
                              f->curr_camera, f->ssr_scale, f->ssao_scale, nullptr, f->ave_log_lum};
    if (!fused) return mifx_composite_execute(ctx, &ca, comp);
    if (chain->fuse_composite_taa && chain->taa->technique_ready)
    {
        chain->pending_composite = ca; // (its image descriptors are the caller's locals and the frame's: alive until the TAA call of this frame, made from the same scope)
        chain->pending_fused     = mifx::TaaFusedComposite{&chain->pending_composite, &ssr->cleanup_in};
        return MIFX_OK;
    }
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    MifxKernelTimer timer(ctx, "composite_ssr_cleanup_kernel");
    const Rows rows = ctx->needed_rows(int(comp->height));
    return launch_composite(ctx->stream, ca, comp, rows.b, rows.e, &ssr->cleanup_in);
}
// TemporalAntiAliasing::Execute on the jittered composite (:871-897) -- on the plane, or with the composite evaluated in place (above)
static mifx_status chain_taa(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* comp)
{
    mifx_taa_render_attribs ta{chain->ctx, comp, f->taa};
    const mifx::TaaFusedComposite fused = chain->pending_fused; // (a per-frame request: cleared whatever the call returns)
    chain->pending_fused = mifx::TaaFusedComposite{nullptr, nullptr};
    chain->taa->fused_composite = fused.attribs ? &fused : nullptr;
    return mifx_taa_execute(chain->taa, &ta);
}

// HnPostProcessTask::Prepare: per-frame PrepareResources in the order PostFX, SSAO, SSR, TAA, Bloom (:671-682), then the chain's own planes
extern "C++" mifx_status mifx::chain_prepare_resources(mifx_chain* chain, const mifx_chain_frame* f)
{
    mifx_postfx* ctx = chain->ctx;
    const uint32_t W = f->frame.Width, H = f->frame.Height;
    MIFX_CHECK(mifx_postfx_prepare(ctx, &f->frame, chain->postfx_flags));
    MIFX_CHECK(mifx_ssao_prepare(chain->ssao, ctx, chain->ssao_flags));
    MIFX_CHECK(mifx_ssr_prepare(chain->ssr, ctx, chain->ssr_flags));
    MIFX_CHECK(mifx_taa_prepare(chain->taa, ctx, f->taa_feature_flags));
    MIFX_CHECK(mifx_bloom_prepare(chain->bloom, ctx, 0));
    MIFX_CHECK(chain->radiance.alloc(W, H, MIFX_FORMAT_F32X4));
    MIFX_CHECK(chain->specular_ibl.alloc(W, H, MIFX_FORMAT_F32X4));
    MIFX_CHECK(chain->composite.alloc(W, H, MIFX_FORMAT_F32X4));
    return MIFX_OK;
}

// Creates the chain's extra streams and events on first use.
extern "C++" mifx_status mifx::chain_make_lanes(mifx_chain* chain, bool three)
{
    if (!chain->side)
    {
        // (a stream priority for the second stream was measured: lowest 1.713-1.723 ms, highest 1.728-1.736 ms, default 1.701-1.703 ms per 4K frame in mode 2)
        MIFX_HIP_CHECK(hipStreamCreateWithFlags(&chain->side, hipStreamNonBlocking));
        for (hipEvent_t* e : {&chain->evFork, &chain->evPrep, &chain->evSsao, &chain->evPrepConsumed}) MIFX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    if (three && !chain->lane_x)
    {
        MIFX_HIP_CHECK(hipStreamCreateWithFlags(&chain->lane_x, hipStreamNonBlocking));
        for (hipEvent_t* e : {&chain->evBloomDone, &chain->evJoinS, &chain->evJoinX, &chain->evXEnd[0], &chain->evXEnd[1], &chain->evSsrDone}) MIFX_HIP_CHECK(hipEventCreateWithFlags(e, hipEventDisableTiming));
    }
    return MIFX_OK;
}

// Whether the lanes of this frame may start behind the previous frame's events alone: the previous frame recorded them, and the library queued nothing on the context
// stream since (history fills of a reset or of a re-allocating prepare, history imports, a new stream: mifx_postfx::stream_epoch).  Otherwise every lane is ordered
// behind the context stream once.
extern "C++" bool mifx::chain_lanes_continue(mifx_chain* chain)
{
    const bool cont = chain->prep_consumed && chain->seen_epoch == chain->ctx->stream_epoch;
    chain->prep_consumed = false; // (set again by a frame that got as far as recording the event: an error return leaves it off)
    return cont;
}

static mifx_status chain_composite(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* radiance, const mifx_image2d* spec, const mifx_image2d* ssao_out,
                                   const mifx_image2d* comp);
static mifx_status chain_taa(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* comp);
static mifx_status chain_bloom_and_tone_map(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d& taa_out, const mifx_image2d* out_ldr, const mifx_native_image* out_native,
                                            hipEvent_t between);

// mifx_chain_set_overlap 3: one frame as three lanes that slide against each other across frames.  The frame's kernels fall into three resource classes --
//   lane S (side stream):    PBR shade (+ R2), PostFX prep, Hi-Z (R1), SSAO A2 .. A8      vector-ALU bound (shade, A3) + their small pyramids
//   lane X (second stream):  SSR R4 .. R6, composite (+ R7), TAA, depth of field          latency / bandwidth bound (the ray march waits 73 % of its cycles)
//   lane M (context stream): Bloom pyramid + final pass (+ tone map)                     ~15 small dependent launches
// -- and a lane only waits for what it reads:  X for the Hi-Z of S (hence shade + prep), the composite for S's SSAO, M for X's TAA / depth of field; the next frame's S for
// this frame's last reader of the planes S overwrites (= the end of X), the next frame's X follows this frame's X in stream order, and before TAA it waits for the
// previous frame's Bloom (the reader of the accumulation buffer / depth-of-field output it is about to overwrite).  So the ray march of frame N runs beside A3 of
// frame N (both need wave slots of every SIMD: the march leaves the ALUs idle, A3 fills them), and the Bloom pyramid of frame N beside the shade of frame N + 1.
// Same kernels, same arguments, same results as one stream; the input contract is that of mode 2 (a frame's input planes are complete when execute is called).
// The context stream ends the frame behind all three lanes: work queued on it after the call sees the whole frame.
//
// mifx_chain_set_overlap 4 (round 5): the same lanes with TWO frames in flight.  In mode 3 lane S of frame N + 1 waits for the end of lane X of frame N, so R5, R6, the
// composite and TAA of every frame -- 0.46 ms of mostly bandwidth-bound work -- run with nothing beside them (profiles/r04_overlap_stats_v17.txt: one kernel in flight
// 67 % of the time).  Here S of frame N + 1 waits for the end of X of frame N - 1 and so runs beside X of frame N.  What S writes and X reads -- radiance, specular IBL,
// SSR's roughness / mask / depth hierarchy, the PostFX planes and the blue noise -- exists twice: the chain trades those planes with its `shadow` set at the start of
// every frame (kernels already queued hold the old addresses by value; the effect objects allocate, fill and hand out "their" planes as always).  Everything else is
// either private to a lane (stream order) or a ping-pong by FrameDesc.Index, which is two deep already -- provided the indices are consecutive: a frame whose index
// does not follow its predecessor's waits for the end of X of the previous frame like mode 3.
static void chain_swap_shadow(mifx_chain* chain)
{
    mifx_postfx* ctx = chain->ctx;
    mifx_ssr*    ssr = chain->ssr;
    mifx_chain::Shadow& sh = chain->shadow;
    chain->radiance.swap(sh.radiance);
    chain->specular_ibl.swap(sh.specular_ibl);
    ctx->reproj_depth.swap(sh.reproj_depth);
    ctx->closest_motion.swap(sh.closest_motion);
    ctx->noise_xy.swap(sh.noise_xy);
    ctx->noise_zw.swap(sh.noise_zw);
    ctx->prev_depth16.swap(sh.prev_depth16);
    ssr->roughness.swap(sh.roughness);
    ssr->mask.swap(sh.mask);
    for (int k = 0; k < mifx_ssr::kMips; ++k) ssr->hiz[k].swap(sh.hiz[k]);
    ssr->hiz_slab.swap(sh.hiz_slab);
}
// the shadow set mirrors the geometry of the live one (after the frame's prepare calls)
static mifx_status chain_prepare_shadow(mifx_chain* chain)
{
    mifx_postfx* ctx = chain->ctx;
    mifx_ssr*    ssr = chain->ssr;
    mifx_chain::Shadow& sh = chain->shadow;
    MIFX_CHECK(sh.radiance.alloc_like(chain->radiance));
    MIFX_CHECK(sh.specular_ibl.alloc_like(chain->specular_ibl));
    MIFX_CHECK(sh.reproj_depth.alloc_like(ctx->reproj_depth));
    MIFX_CHECK(sh.closest_motion.alloc_like(ctx->closest_motion));
    MIFX_CHECK(sh.noise_xy.alloc_like(ctx->noise_xy));
    MIFX_CHECK(sh.noise_zw.alloc_like(ctx->noise_zw));
    MIFX_CHECK(sh.prev_depth16.alloc_like(ctx->prev_depth16));
    MIFX_CHECK(sh.roughness.alloc_like(ssr->roughness));
    MIFX_CHECK(sh.mask.alloc_like(ssr->mask));
    if (sh.hiz[0].data == nullptr || sh.hiz[0].w != ssr->hiz[0].w || sh.hiz[0].h != ssr->hiz[0].h) MIFX_CHECK(mifx::ssr_alloc_hiz(ssr->w, ssr->h, sh.hiz, sh.hiz_slab));
    return MIFX_OK;
}

// MIFX_LANE_EDGES / mifx_chain_set_lane_edges: at the launch site of a named kernel (mifx_postfx::kernel_hook, called by MifxKernelTimer on the lane's stream)
//   end:   record "kernel `name` of frame seq is done" when an edge names it as a signal
//   begin: wait for the signals of the edges that name it as the waiter, if that frame recorded them
static void chain_kernel_hook(mifx_chain* chain, const char* name, bool begin)
{
    const hipStream_t s = chain->ctx->stream;
    for (const mifx_chain::Edge& e : chain->edges)
    {
        if (begin && e.waiter == name)
        {
            auto it = chain->signals.find(e.signal);
            if (it == chain->signals.end() || chain->seq < uint64_t(e.delta)) continue;
            const uint64_t want = chain->seq - uint64_t(e.delta);
            if (it->second.seq[want & 3u] == want) (void)hipStreamWaitEvent(s, it->second.ev[want & 3u], 0);
        }
        else if (!begin && e.signal == name)
        {
            mifx_chain::Signal& sg = chain->signals[e.signal];
            hipEvent_t& ev = sg.ev[chain->seq & 3u];
            if (!ev && hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) continue;
            if (sg.seq[chain->seq & 3u] == chain->seq) continue; // (one record per frame: the first launch of that name)
            if (hipEventRecord(ev, s) == hipSuccess) sg.seq[chain->seq & 3u] = chain->seq;
        }
    }
}

// mifx_chain_set_overlap 5 (round 6): mode 4 with the frame's bandwidth-bound tail -- the composite, TAA, depth of field -- on lane M in front of Bloom instead of at the end
// of lane X.  In mode 4 the march of frame N + 1 follows TAA of frame N in stream order, so the composite and TAA still run with little beside them (profiles/
// r06_overlap_stats_v11.txt: 49 % and 42 % of their time alone); here lane X of a frame is R4 .. R6 only, and the next frame's march runs beside this frame's composite and TAA:
//   S  shade, prep, Hi-Z, SSAO   |   X  R4, R5, R6   |   M  composite (+ R7), TAA, depth of field, Bloom, tone map        (0.67 / 0.52 / 0.51 ms of kernels at 4K)
// What the composite and TAA read of lanes S and X is either double-buffered already (mode 4's shadow set: radiance, specular IBL, roughness, mask, the PostFX planes) or a
// ping-pong by FrameDesc.Index (SSR's and SSAO's histories: frame N + 1 writes the other slot, frame N + 2 -- whose lane S waits for this frame's TAA -- the same one).
static mifx_status chain_execute_lanes(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr, const mifx_native_image* out_native, int mode)
{
    const bool pipelined = mode >= 4, late = mode >= 5;
    mifx_postfx*      ctx = chain->ctx;
    const hipStream_t M   = ctx->stream;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    MIFX_CHECK(chain_make_lanes(chain, true));
    const hipStream_t S = chain->side, X = chain->lane_x;
    if (pipelined)
    {
        MIFX_CHECK(chain_prepare_shadow(chain));
        chain_swap_shadow(chain); // this frame's set = the one the frame before the previous frame used
    }
    const mifx_image2d radiance = chain->radiance.desc(), spec = chain->specular_ibl.desc(), comp = chain->composite.desc();
    struct Rejoin // whatever happens, the context stream ends behind both lanes and is the context's stream again
    {
        mifx_chain* c;
        hipStream_t m;
        bool        done = false;
        ~Rejoin()
        {
            c->ctx->stream      = m;
            c->ctx->kernel_hook = nullptr;
            if (done) return; // (the regular path joined through evSsao -> evPrepConsumed)
            if (hipEventRecord(c->evJoinS, c->side) == hipSuccess) (void)hipStreamWaitEvent(m, c->evJoinS, 0);
            if (hipEventRecord(c->evJoinX, c->lane_x) == hipSuccess) (void)hipStreamWaitEvent(m, c->evJoinX, 0);
        }
    } rejoin{chain, M};
    const uint64_t k = chain->seq;
    if (chain_lanes_continue(chain))
    {
        // lane S behind the last reader of what it overwrites: the end of lane X of the previous frame -- or, two frames in flight, of the frame before that one
        const bool deep = pipelined && k >= 2 && f->frame.Index == chain->last_index + 1u;
        MIFX_HIP_CHECK(hipStreamWaitEvent(S, deep ? chain->evXEnd[k & 1u] : chain->evPrepConsumed, 0)); // (evXEnd[k & 1] still holds frame k - 2's record)
    }
    else
    {
        MIFX_HIP_CHECK(hipEventRecord(chain->evFork, M));
        MIFX_HIP_CHECK(hipStreamWaitEvent(S, chain->evFork, 0));
        MIFX_HIP_CHECK(hipStreamWaitEvent(X, chain->evFork, 0));
    }
    if (pipelined && !chain->edges.empty()) ctx->kernel_hook = [chain](const char* name, bool begin) { chain_kernel_hook(chain, name, begin); };
    mifx_postfx_render_attribs pa{f->gbuffer.depth, f->prev_depth, f->motion, f->curr_camera, f->prev_camera};
    mifx_ssr_render_attribs    sr{ctx, &radiance, f->gbuffer.depth, f->gbuffer.normal, f->gbuffer.material, f->motion, f->ssr};
    mifx_ssao_render_attribs   sa{ctx, f->gbuffer.depth, f->gbuffer.normal, f->ssao};
    // lane S: shade, prep
    ctx->stream = S;
    MIFX_CHECK(chain_shade(chain, f, &radiance, &spec));
    MIFX_CHECK(mifx_postfx_execute(ctx, &pa));
    // lane X: SSR, its depth hierarchy on S (X waits for evPrep behind it, i.e. for the shade, the prep pass and the hierarchy)
    // (round 5, measured and not kept: the prep pass and the hierarchy -- 68 us of streaming over the inputs -- on a fourth stream beside the shade, as the sharded frame does
    //  with its whole-frame hierarchy: 1.6358 against 1.6418 ms over three alternating pairs of runs on one box, inside the noise; order-checked by tests/cpu_product/order.py)
    ctx->stream = X;
    // (mode 5 with R7 as a pass of its own -- fusion bit 2 off: R7 of this frame overwrites the SSR output plane the PREVIOUS frame's composite, now on lane M, may still be
    //  reading; found by tests/cpu_product/order.py.  That configuration waits for the previous frame's composite / TAA here; with R7 inside the composite there is no such plane)
    if (late && !chain->fuse_ssr_cleanup) MIFX_HIP_CHECK(hipStreamWaitEvent(X, chain->evPrepConsumed, 0));
    chain->ssr->defer_cleanup = chain->fuse_ssr_cleanup;
    chain->ssr->hiz_stream    = S;
    chain->ssr->hiz_done      = chain->evPrep;
    MIFX_CHECK(mifx_ssr_execute(chain->ssr, &sr));
    // lane S: SSAO
    ctx->stream = S;
    MIFX_CHECK(mifx_ssao_execute(chain->ssao, &sa));
    MIFX_HIP_CHECK(hipEventRecord(chain->evSsao, S));
    // lane X (mode 5: lane M, behind this frame's SSR): composite, TAA, depth of field
    const hipStream_t T = late ? M : X;
    if (late)
    {
        MIFX_HIP_CHECK(hipEventRecord(chain->evSsrDone, X));
        MIFX_HIP_CHECK(hipStreamWaitEvent(M, chain->evSsrDone, 0));
    }
    ctx->stream = T;
    MIFX_HIP_CHECK(hipStreamWaitEvent(T, chain->evSsao, 0));
    mifx_image2d ssao_out, taa_out;
    MIFX_CHECK(mifx_ssao_get_output(chain->ssao, &ssao_out));
    MIFX_CHECK(chain_composite(chain, f, &radiance, &spec, &ssao_out, &comp));
    if (!late) MIFX_HIP_CHECK(hipStreamWaitEvent(X, chain->evBloomDone, 0)); // (recorded by the previous frame; never recorded = no wait.  Mode 5: the previous Bloom is earlier on this stream)
    MIFX_CHECK(chain_taa(chain, f, &comp));
    MIFX_CHECK(mifx_taa_get_output(chain->taa, 0, &taa_out));
    if (chain->dof)
    {
        MIFX_CHECK(mifx_dof_prepare(chain->dof, ctx, chain->dof_flags));
        mifx_dof_render_attribs da{ctx, &taa_out, f->gbuffer.depth, &chain->dof_attribs};
        MIFX_CHECK(mifx_dof_execute(chain->dof, &da));
        MIFX_CHECK(mifx_dof_get_output(chain->dof, &taa_out));
    }
    MIFX_HIP_CHECK(hipEventRecord(chain->evPrepConsumed, T));
    if (pipelined) MIFX_HIP_CHECK(hipEventRecord(chain->evXEnd[k & 1u], T));
    // lane M: Bloom, tone map
    ctx->stream = M;
    if (!late) MIFX_HIP_CHECK(hipStreamWaitEvent(M, chain->evPrepConsumed, 0));
    rejoin.done = true;
    MIFX_CHECK(chain_bloom_and_tone_map(chain, f, taa_out, out_ldr, out_native, nullptr));
    MIFX_HIP_CHECK(hipEventRecord(chain->evBloomDone, M));
    chain->seen_epoch    = ctx->stream_epoch;
    chain->prep_consumed = true;
    chain->timed         = false;
    chain->last_index    = f->frame.Index;
    if (pipelined) ++chain->seq;
    return MIFX_OK;
}

static mifx_status chain_execute_impl(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr, const mifx_native_image* out_native)
{
    MIFX_REQUIRE(chain != nullptr && f != nullptr && (out_ldr != nullptr) != (out_native != nullptr), "mifx_chain_execute: null argument");
    MIFX_REQUIRE(f->curr_camera && f->prev_camera && f->ibl && f->pbr && f->ssao && f->ssr && f->taa && f->bloom && f->tone_mapping,
                 "mifx_chain_execute: every attribs pointer of mifx_chain_frame must be set");
    MIFX_CHECK(chain_check_workflow(f));
    mifx_postfx* ctx = chain->ctx;
    MIFX_CHECK(mifx::chain_prepare_resources(chain, f));
    if (chain->overlap >= 3 && !chain->profiling) return chain_execute_lanes(chain, f, out_ldr, out_native, chain->overlap);
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
        MIFX_CHECK(chain_make_lanes(chain, false));
        // Across frames (overlap 2; the caller guarantees that the frame's input planes are complete when execute is called): the side stream does not wait for the
        // previous frame's Bloom and tone map, only for its last reader of what prep and SSAO overwrite (the PostFX planes and the blue noise: SSR, TAA, depth of field),
        // so that the next frame's prep + SSAO fill the GPU under the small launches of the Bloom pyramid.
        // (Measured and dropped: the PostFX planes double-buffered so that the second stream may run a whole frame ahead, beside this frame's ray march and TAA
        //  instead of its Bloom: 1.744 vs 1.742 ms; Bloom and the tone map of the frame on the second stream as well, so that both streams are busy all the time:
        //  1.706 / 1.720 vs 1.708 / 1.704 ms.  The gain of running two kernels at once saturates at ~5.6 %, which is also what two whole chains on two streams reach
        //  (tools/exp_two_chains.py).  tools/overlap_stats.py shows what runs beside what: profiles/r03_overlap_stats.txt.)
        // (only while nothing else was queued on the context stream in between -- a reset's or a re-allocating prepare's history fills, an import: chain_lanes_continue)
        if (chain_lanes_continue(chain) && chain->overlap >= 2) MIFX_HIP_CHECK(hipStreamWaitEvent(chain->side, chain->evPrepConsumed, 0));
        else
        {
            MIFX_HIP_CHECK(hipEventRecord(chain->evFork, main));
            MIFX_HIP_CHECK(hipStreamWaitEvent(chain->side, chain->evFork, 0));
        }
        ctx->stream = chain->side;
        mifx_status st = mifx_postfx_execute(ctx, &pa);
        if (st >= 0) st = hipEventRecord(chain->evPrep, chain->side) == hipSuccess ? MIFX_OK : MIFX_ERR_HIP;
        if (st >= 0) st = mifx_ssao_execute(chain->ssao, &sa);
        if (st >= 0) st = hipEventRecord(chain->evSsao, chain->side) == hipSuccess ? MIFX_OK : MIFX_ERR_HIP;
        ctx->stream = main;
        MIFX_CHECK(st);
        MIFX_CHECK(chain_shade(chain, f, &radiance, &spec)); // (measured: the shade on the side stream as well, in front of prep, changes nothing: 1.747 vs 1.751 ms)
        MIFX_HIP_CHECK(hipStreamWaitEvent(main, chain->evPrep, 0));
        chain->ssr->defer_cleanup = chain->fuse_ssr_cleanup;
        MIFX_CHECK(mifx_ssr_execute(chain->ssr, &sr));
        MIFX_HIP_CHECK(hipStreamWaitEvent(main, chain->evSsao, 0));
        stage += 4;
    }
    else
    {
        // forward shade (stands in for HnRenderRprimsTask: SceneColor + the IBL target of the USD G-buffer)
        MIFX_CHECK(chain_shade(chain, f, &radiance, &spec));
        MIFX_CHECK(mark());
        // PostFXContext::Execute (:788-809)
        MIFX_CHECK(mifx_postfx_execute(ctx, &pa));
        MIFX_CHECK(mark());
        // ScreenSpaceReflection::Execute (:811-822)
        chain->ssr->defer_cleanup = chain->fuse_ssr_cleanup;
        MIFX_CHECK(mifx_ssr_execute(chain->ssr, &sr));
        MIFX_CHECK(mark());
        // ScreenSpaceAmbientOcclusion::Execute (:824-832)
        MIFX_CHECK(mifx_ssao_execute(chain->ssao, &sa));
        MIFX_CHECK(mark());
    }
    mifx_image2d ssao_out, taa_out, bloom_out;
    MIFX_CHECK(mifx_ssao_get_output(chain->ssao, &ssao_out));
    // composite draw (:834-869), no tone mapping while TAA is on
    MIFX_CHECK(chain_composite(chain, f, &radiance, &spec, &ssao_out, &comp));
    MIFX_CHECK(mark());
    // TemporalAntiAliasing::Execute on the jittered composite (:871-897)
    MIFX_CHECK(chain_taa(chain, f, &comp));
    MIFX_CHECK(mifx_taa_get_output(chain->taa, 0, &taa_out));
    MIFX_CHECK(mark());
    // DepthOfField::Execute on the TAA output (:899-909; m_UseDOF requires TAA, :654)
    if (chain->dof)
    {
        MIFX_CHECK(mifx_dof_prepare(chain->dof, ctx, chain->dof_flags));
        mifx_dof_render_attribs da{ctx, &taa_out, f->gbuffer.depth, &chain->dof_attribs};
        MIFX_CHECK(mifx_dof_execute(chain->dof, &da));
        MIFX_CHECK(mifx_dof_get_output(chain->dof, &taa_out));
    }
    chain->prep_consumed = false;
    if (chain->overlap >= 2 && !chain->profiling && chain->evPrepConsumed)
    {
        MIFX_HIP_CHECK(hipEventRecord(chain->evPrepConsumed, ctx->stream));
        chain->seen_epoch    = ctx->stream_epoch;
        chain->prep_consumed = true;
    }
    MIFX_CHECK(mark());
    MIFX_CHECK(chain_bloom_and_tone_map(chain, f, taa_out, out_ldr, out_native, chain->profiling ? chain->ev[stage] : nullptr)); // (the stage mark between the two)
    ++stage;
    MIFX_CHECK(mark());
    chain->timed = chain->profiling;
    return MIFX_OK;
}

// Bloom::Execute on the TAA (or depth-of-field) output (HnPostProcessTask.cpp:911-918) and the copy-frame draw = ToneMap (+ sRGB) (:920-926)
static mifx_status chain_bloom_and_tone_map(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d& taa_out, const mifx_image2d* out_ldr, const mifx_native_image* out_native,
                                            hipEvent_t between)
{
    mifx_postfx* ctx = chain->ctx;
    mifx_image2d bloom_out;
    mifx_bloom_render_attribs ba{ctx, &taa_out, f->bloom};
    // With a plain fp32 target and a constant average luminance the copy-frame ToneMap() (:920-926) is the tail of Bloom's final up-sample: the Bloom output
    // is written as always, the LDR frame in the same pass (bit-identical to the two passes; the "tonemap" stage time is then part of "bloom").
    const bool fuse_tone_map = out_native == nullptr && chain->auto_exposure == nullptr && chain->fuse_tone_map;
    const mifx_bloom::FusedToneMap ftm{out_ldr, f->tone_mapping, f->ave_log_lum, f->tonemap_flags, chain->fuse_bloom_output};
    MIFX_REQUIRE(chain->bloom->prepared, "mifx_chain_execute: bloom resources are not prepared");
    MIFX_CHECK(chain->bloom->run(&ba, 0, fuse_tone_map ? &ftm : nullptr));
    if (between) MIFX_HIP_CHECK(hipEventRecord(between, ctx->stream));
    if (fuse_tone_map) return MIFX_OK; // (the frame is written; the Bloom output plane on demand, mifx_bloom::run_deferred_output)
    MIFX_CHECK(mifx_bloom_get_output(chain->bloom, &bloom_out));
    // copy-frame draw = ToneMap (+ sRGB) (:920-926); with auto exposure on, fAveLogLum is the adapted average luminance of the scene colour
    if (out_native != nullptr)
    {
        // the copy-frame target in its own format (the swap chain's in the reference): the conversion is the tail of the tone-map kernel
        MIFX_REQUIRE(chain->auto_exposure == nullptr, "mifx_chain_execute_native: not combined with auto exposure");
        MIFX_CHECK(mifx_tonemap_execute_native(ctx, &bloom_out, out_native, f->tone_mapping, f->ave_log_lum, f->tonemap_flags));
    }
    else if (chain->auto_exposure)
    {
        MIFX_CHECK(mifx_autoexposure_execute(chain->auto_exposure, &bloom_out, chain->ae_elapsed, chain->ae_adapt ? 1 : 0));
        MIFX_CHECK(mifx_tonemap_execute_auto(ctx, &bloom_out, out_ldr, f->tone_mapping, chain->auto_exposure, f->tonemap_flags));
    }
    else
        MIFX_CHECK(mifx_tonemap_execute(ctx, &bloom_out, out_ldr, f->tone_mapping, f->ave_log_lum, f->tonemap_flags));
    return MIFX_OK;
}

extern "C" mifx_status mifx_chain_execute(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr)
{
    MIFX_REQUIRE(out_ldr != nullptr, "mifx_chain_execute: null output");
    if (chain != nullptr && (chain->halo_ssao_pending || chain->halo_rest_pending)) chain->join_halos(); // (an unsharded frame after sharded ones writes whole history planes)
    return chain_execute_impl(chain, f, out_ldr, nullptr);
}
extern "C" mifx_status mifx_chain_execute_native(mifx_chain* chain, const mifx_chain_frame* f, const mifx_native_image* out_native)
{
    MIFX_REQUIRE(out_native != nullptr, "mifx_chain_execute_native: null output");
    if (chain != nullptr && (chain->halo_ssao_pending || chain->halo_rest_pending)) chain->join_halos();
    return chain_execute_impl(chain, f, nullptr, out_native);
}

// ------------------------------------------------------------------------------------------------ row-band sharding
namespace
{
struct ShardRows
{
    Rows band, taa, comp, prep; // rows of the final image owned; rows TAA / composite (= shade, SSR, SSAO outputs) / PostFX prep are computed on
    Rows post;                  // rows of Bloom's input (the TAA output, or the depth-of-field output computed from it) that Bloom and the final pass read
    Rows need;                  // rows of the Bloom output computed: the band, plus one row either side when auto exposure samples it (bilinear footprints
                                // of the low-resolution luminance rows this rank writes, mifx_autoexposure::sample_rows)
};
// Reaches: Bloom's fine levels read the TAA output on mifx_bloom::Plan::taa; TAA reads the 3x3 neighbourhood of the composite; SSAO and SSR
// derive their internal windows from the rows of their output (api_ssao.cpp, api_ssr.cpp) and need the prep outputs on the largest of them
// (<= 1 + radius 4 + 1 + 48 + 15 alignment (mifx_ssao::kWindowAlign; 31 until round 5) + 1 rows beyond the composite rows: 96 is checked by both effects against prep_rows).
} // namespace
extern "C++" bool mifx::shard_bloom_halo_enabled() // MIFX_SHARD_BLOOM_HALO=0: round 5's frame (every rank produces the level-0 rows it reads)
{
    // (read per frame, not once: tests/test_gpu_sharded.py holds mifx_chain_execute_band to the phases driven one by one, which have no such exchange, in the same process
    //  as the tests of the exchange)
    const char* e = std::getenv("MIFX_SHARD_BLOOM_HALO");
    return e == nullptr || std::atoi(e) != 0;
}
namespace
{
ShardRows shard_rows(const mifx_chain* chain, const mifx_chain_frame* f, Rows band)
{
    const int H = int(f->frame.Height);
    ShardRows r;
    r.band = rows_clip(band, H);
    r.need = chain->auto_exposure ? rows_expand(r.band, 1, H) : r.band;
    const mifx_bloom::Plan p = chain->bloom->make_plan(r.band, r.need, chain->bloom->mip_count(*f->bloom));
    r.post = p.G >= 0 ? rows_hull(p.taa, r.need) : Rows{0, H};
    r.taa  = r.post;
    // depth of field sits between TAA and Bloom: its output on r.post needs the TAA output on the colour rows of mifx_dof::windows (bokeh gather and fill radii)
    if (chain->dof) r.taa = mifx_dof::windows(chain->dof_attribs, r.post, int(f->frame.Width), H).colour;
    r.comp = rows_expand(r.taa, 1, H);
    r.prep = rows_expand(r.comp, 96, H);
    // (its colour-independent passes -- CoC, temporal CoC, dilation, blur -- run whole on every rank: the temporal CoC reads the closest motion vectors of the whole frame)
    if (chain->dof && (chain->dof_flags & MIFX_DOF_FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING)) r.prep = Rows{0, H};
    return r;
}
} // namespace

extern "C" mifx_status mifx_chain_set_row_band(mifx_chain* chain, int32_t row_begin, int32_t row_end, int32_t max_motion_rows)
{
    MIFX_REQUIRE(chain != nullptr && row_begin >= 0 && row_end >= row_begin && max_motion_rows >= 0, "mifx_chain_set_row_band: bad argument");
    chain->band       = Rows{row_begin, row_end}; // {0, 0} switches sharding off
    chain->max_motion = max_motion_rows;
    chain->ctx->band  = chain->band;
    chain->ctx->max_motion = max_motion_rows;
    if (chain->band.empty()) chain->ctx->need = Rows{0, 0};
    return MIFX_OK;
}

// One frame in four phases; between them the caller exchanges planes with the other ranks (diligentfx_amd/sharded.py: ShardedChain):
//   phase 0: PBR shade on the composite rows
//   phase 1: PostFX prep, SSAO
//   phase 2: SSR, composite, TAA, Bloom fine levels        -> "bloom_gather": every rank contributes the rows it owns, all ranks get the level
//            (the rays of SSR hit anywhere in the frame: the colour at a hit outside the rows of phase 0 is shaded on the spot -- no radiance exchange)
//   phase 3: Bloom coarse levels + up-sampling, tone map   -> halo exchange of the five history planes for the next frame
// With auto exposure on, phase 3 ends with this rank's rows of the low-resolution luminance ("ae_low_res", rows ae_begin..ae_end of mifx_shard_info) instead of
// the tone map; those rows are all-gathered and
//   phase 4: luminance reduction + adaptation (the same 64x64 values in the same order on every rank: the average is bit-identical), tone map
// follows (without auto exposure phase 4 does nothing).
// (SSAO before SSR: the reference runs SSR first, but the two effects only share read-only inputs.)
extern "C" mifx_status mifx_chain_execute_phase(mifx_chain* chain, const mifx_chain_frame* f, const mifx_image2d* out_ldr, int32_t phase)
{
    MIFX_REQUIRE(chain != nullptr && f != nullptr && out_ldr != nullptr && phase >= 0 && phase <= 4, "mifx_chain_execute_phase: bad argument");
    MIFX_REQUIRE(!chain->band.empty(), "mifx_chain_execute_phase: no row band set (mifx_chain_set_row_band)");
    MIFX_REQUIRE(f->curr_camera && f->prev_camera && f->ibl && f->pbr && f->ssao && f->ssr && f->taa && f->bloom && f->tone_mapping,
                 "mifx_chain_execute_phase: every attribs pointer of mifx_chain_frame must be set");
    MIFX_CHECK(chain_check_workflow(f));
    mifx_postfx* ctx = chain->ctx;
    const uint32_t W = f->frame.Width, H = f->frame.Height;
    if (phase == 0) MIFX_CHECK(mifx::chain_prepare_resources(chain, f));
    const mifx_image2d radiance = chain->radiance.desc(), spec = chain->specular_ibl.desc(), comp = chain->composite.desc();
    const ShardRows r = shard_rows(chain, f, chain->band);
    struct NeedGuard // the row request is per call: never leave one behind for a later whole-frame call
    {
        mifx_postfx* c;
        ~NeedGuard() { c->need = Rows{0, 0}; }
    } guard{ctx};
    if (phase == 0)
    {
        ctx->need = r.comp;
        return chain_shade(chain, f, &radiance, &spec);
    }
    mifx_image2d ssao_out, taa_out, bloom_out;
    mifx_bloom_render_attribs ba{ctx, nullptr, f->bloom};
    // A history-halo exchange of the previous sharded frame that is still travelling (mifx_chain_execute_sharded, async_halos) is joined where this frame first reads the
    // plane -- also for a caller that mixes mifx_chain_execute_sharded frames with phases of its own.
    if (phase == 1 && chain->halo_ssao_pending)
    {
        MIFX_HIP_CHECK(hipStreamWaitEvent(ctx->stream, chain->evHaloSsao, 0)); // A5 reprojects into the ghost rows of last frame's AO / history length
        chain->halo_ssao_pending = false;
    }
    if (phase == 2 && chain->halo_rest_pending)
    {
        MIFX_HIP_CHECK(hipStreamWaitEvent(ctx->stream, chain->evHaloRest, 0)); // R6 and TAA reproject into the ghost rows of last frame's SSR / TAA histories
        chain->halo_rest_pending = false;
    }
    if (phase == 1)
    {
        mifx_postfx_render_attribs pa{f->gbuffer.depth, f->prev_depth, f->motion, f->curr_camera, f->prev_camera};
        ctx->need = r.prep;
        MIFX_CHECK(mifx_postfx_execute(ctx, &pa));
        if (chain->sig_after_prep) MIFX_HIP_CHECK(hipEventRecord(chain->sig_after_prep, ctx->stream)); // (the sharded frame's SSAO lane: SSR on the other lane waits for this)
        ctx->need = r.comp;
        mifx_ssao_render_attribs sa{ctx, f->gbuffer.depth, f->gbuffer.normal, f->ssao};
        return mifx_ssao_execute(chain->ssao, &sa);
    }
    if (phase == 2)
    {
        ctx->need = r.comp;
        mifx_ssr_render_attribs sr{ctx, &radiance, f->gbuffer.depth, f->gbuffer.normal, f->gbuffer.material, f->motion, f->ssr};
        chain->ssr->defer_cleanup = chain->fuse_ssr_cleanup;
        // No exchange of the shaded radiance: the ray march records where every ray hit, and the hit fetch loads the colour from the rows this rank shaded in
        // phase 0 or shades the hit pixel itself (the G-buffer and the IBL maps are whole on every rank; same kernel body, bit-identical colour).
        if (chain->shaded_rows.empty() || chain->shaded_frame != f->frame.Index)
        {
            set_error("mifx_chain_execute_phase: phase 2 of frame %u before its phase 0 (the hit fetch needs the rows that phase shaded)", f->frame.Index);
            return MIFX_ERR_INVALID_OP;
        }
        chain->ssr->after_trace = [chain, f, ctx, radiance](Img rays, Img coords) -> mifx_status {
            MIFX_HIP_CHECK(hipSetDevice(ctx->device));
            MifxKernelTimer timer(ctx, "pbr_hit_fetch_kernel");
            if (chain->has_layers || chain->has_shadows) // the frame was shaded with mifx_chain_set_material_layers: the hit pixels take the same permutation
            {
                const mifx_pbr_layers none{};
                const LayeredHitFetch hit{rays, coords, chain->shaded_rows.b, chain->shaded_rows.e};
                return launch_pbr_shade_layers(ctx->stream, ctx->ibl_apron, &f->gbuffer, chain->has_layers ? chain->layers : none, *f->curr_camera, *f->pbr, f->ibl, f->background, &radiance,
                                               nullptr, 0, 0, (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0, chain->has_shadows ? &chain->shadows : nullptr, &hit);
            }
            return launch_pbr_hit_fetch(ctx->stream, ctx->ibl_apron, &f->gbuffer, *f->curr_camera, *f->pbr, f->ibl, f->background, rays, coords, &radiance, chain->shaded_rows.b,
                                        chain->shaded_rows.e, (ctx->flags & MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH) != 0);
        };
        chain->ssr->hit_local_rows = chain->shaded_rows; // (R4 loads the colour of a hit in these rows itself; only the others go through the fetch)
        const mifx_status st_ssr = mifx_ssr_execute(chain->ssr, &sr);
        chain->ssr->after_trace    = nullptr;
        chain->ssr->hit_local_rows = Rows{0, 0};
        MIFX_CHECK(st_ssr);
        if (chain->wait_before_composite) MIFX_HIP_CHECK(hipStreamWaitEvent(ctx->stream, chain->wait_before_composite, 0)); // (the end of SSAO on its lane)
        MIFX_CHECK(mifx_ssao_get_output(chain->ssao, &ssao_out));
        MIFX_CHECK(chain_composite(chain, f, &radiance, &spec, &ssao_out, &comp));
        ctx->need = r.taa;
        MIFX_CHECK(chain_taa(chain, f, &comp));
        MIFX_CHECK(mifx_taa_get_output(chain->taa, 0, &taa_out));
        if (chain->dof) // DepthOfField::Execute on the TAA output, on the rows Bloom reads of it
        {
            ctx->need = r.post;
            MIFX_CHECK(mifx_dof_prepare(chain->dof, ctx, chain->dof_flags));
            mifx_dof_render_attribs da{ctx, &taa_out, f->gbuffer.depth, &chain->dof_attribs};
            MIFX_CHECK(mifx_dof_execute(chain->dof, &da));
            MIFX_CHECK(mifx_dof_get_output(chain->dof, &taa_out));
        }
        ctx->need = r.need;
        ba.color  = &taa_out;
        return chain->bloom->run(&ba, 1);
    }
    mifx_autoexposure* ae = chain->auto_exposure;
    if (phase == 4)
    {
        if (!ae) return MIFX_OK;
        MIFX_HIP_CHECK(hipSetDevice(ctx->device));
        MIFX_CHECK(launch_autoexposure_reduce(ctx->stream, ae->low_res.view(), static_cast<float*>(ae->average.data), chain->ae_elapsed, chain->ae_adapt ? 1 : 0));
        MIFX_CHECK(mifx_bloom_get_output(chain->bloom, &bloom_out));
        ctx->need = r.band;
        return mifx_tonemap_execute_auto(ctx, &bloom_out, out_ldr, f->tone_mapping, ae, f->tonemap_flags);
    }
    MIFX_CHECK(mifx_taa_get_output(chain->taa, 0, &taa_out));
    if (chain->dof) MIFX_CHECK(mifx_dof_get_output(chain->dof, &taa_out)); // (computed in phase 2)
    ctx->need = r.need;
    ba.color  = &taa_out;
    const bool fuse_tone_map = chain->fuse_tone_map && ae == nullptr;
    const mifx_bloom::FusedToneMap ftm{out_ldr, f->tone_mapping, f->ave_log_lum, f->tonemap_flags, chain->fuse_bloom_output};
    MIFX_CHECK(chain->bloom->run(&ba, 2, fuse_tone_map ? &ftm : nullptr));
    if (fuse_tone_map) return MIFX_OK;
    MIFX_CHECK(mifx_bloom_get_output(chain->bloom, &bloom_out));
    if (ae)
    {
        Img  color;
        bool packed = false;
        MIFX_CHECK(to_img_hdr(&bloom_out, "bloom output", color, packed));
        const Rows rows = mifx_autoexposure::sample_rows(r.band, int(H));
        return launch_autoexposure_rows(ctx->stream, color, ae->low_res.view(), rows.b, rows.e, packed);
    }
    return mifx_tonemap_execute(ctx, &bloom_out, out_ldr, f->tone_mapping, f->ave_log_lum, f->tonemap_flags);
}

// What the caller has to move between the phases: planes (valid after the phase that writes them) and row counts.
extern "C" mifx_status mifx_chain_get_shard_info(mifx_chain* chain, const mifx_chain_frame* f, mifx_shard_info* out)
{
    MIFX_REQUIRE(chain != nullptr && f != nullptr && out != nullptr && f->bloom && f->ssao && f->ssr, "mifx_chain_get_shard_info: null argument");
    MIFX_REQUIRE(!chain->band.empty() && chain->bloom->prepared, "mifx_chain_get_shard_info: set a row band and run phase 0 first");
    // (a chain whose frames run through mifx_chain_execute_sharded exchanges Bloom's level-0 halos and so needs shorter history halos: report what that frame uses)
    const bool was = chain->bloom->halo_level0;
    chain->bloom->halo_level0 = chain->comm != nullptr && mifx::shard_bloom_halo_enabled();
    *out = mifx::chain_shard_info(chain, f, chain->band);
    chain->bloom->halo_level0 = was;
    return MIFX_OK;
}

// Bloom's row plan of any band of the frame (mifx_chain_execute_sharded: which rows of level 0 a rank produces and which it reads, for the halo exchange of that level)
extern "C++" mifx_bloom::Plan mifx::chain_bloom_plan(const mifx_chain* chain, const mifx_chain_frame* f, Rows band)
{
    const ShardRows r = shard_rows(chain, f, band);
    return chain->bloom->make_plan(r.band, r.need, chain->bloom->mip_count(*f->bloom));
}

// the same for any band of the frame: the rows a rank owning `band` has to receive (mifx_chain_execute_sharded derives every rank's needs from
// the cuts, so that both sides of an exchange move the same rows without a round of communication)
extern "C++" mifx_shard_info mifx::chain_shard_info(const mifx_chain* chain, const mifx_chain_frame* f, Rows band)
{
    mifx_shard_info info{};
    mifx_shard_info* out = &info;
    {
    const int H = int(f->frame.Height);
    const ShardRows r = shard_rows(chain, f, band);
    const mifx_bloom::Plan p = chain->bloom->make_plan(r.band, r.band, chain->bloom->mip_count(*f->bloom));
    const int m = chain->max_motion;
    auto ghost = [&](Rows w) { const int lo = r.band.b - w.b, hi = w.e - r.band.e; return lo > hi ? lo : hi; };
    // rows a pass reads of its history = its row window grown by the reprojection reach and the filter support
    const bool centredTaps = int(f->curr_camera->f4ViewportSize[0]) == int(f->frame.Width) && int(f->curr_camera->f4ViewportSize[1]) == H && f->frame.Width % 16u == 0u &&
                             f->frame.Height % 16u == 0u && !(chain->ssao_flags & MIFX_SSAO_FEATURE_FLAG_HALF_RESOLUTION); // as in mifx_ssao_execute
    const Rows ssao5 = rows_align(rows_expand(rows_expand(r.comp, int(std::ceil(f->ssao->SpatialReconstructionRadius)) + 1, H), centredTaps ? 24 : 48, H), mifx_ssao::kWindowAlign, H);
    const Rows ssr6  = rows_expand(r.comp, 3, H);
    out->band_begin = r.band.b; out->band_end = r.band.e;
    out->halo_taa   = ghost(r.taa) + m + 3;   // Catmull-Rom history taps: +-2 texels around the reprojected position
    out->halo_ssr   = ghost(ssr6) + 2 * m + 2; // incident and hit-point reprojection, bilinear
    out->halo_ssao  = ghost(ssao5) + m + 2;
    out->gather_level = p.G;
    out->own_begin = p.own.b; out->own_end = p.own.e;
    const Rows ae = chain->auto_exposure ? mifx_autoexposure::sample_rows(r.band, H) : Rows{0, 0};
    out->ae_begin = ae.b; out->ae_end = ae.e;
    }
    return info;
}

extern "C" mifx_status mifx_chain_get_shard_plane(mifx_chain* chain, const char* name, mifx_image2d* out)
{
    MIFX_REQUIRE(chain != nullptr && name != nullptr && out != nullptr, "mifx_chain_get_shard_plane: null argument");
    chain->join_halos(); // (what the caller queues on the context's stream next sees the planes with their halos received)
    const std::string n = name;
    const uint32_t ci = chain->ctx->frame.Index & 1u; // the planes the last executed frame wrote
    const Plane* p = nullptr;
    if (n == "radiance") p = &chain->radiance;
    else if (n == "bloom_gather") p = chain->bloom->prepared && int(chain->bloom->down.size()) > mifx_bloom::kGatherLevel ? chain->bloom->down[mifx_bloom::kGatherLevel] : nullptr;
    else if (n == "taa_history") p = &chain->taa->accum[ci];
    else if (n == "ssr_history_radiance") p = &chain->ssr->hist_radiance[ci];
    else if (n == "ssr_history_variance") p = &chain->ssr->hist_variance[ci];
    else if (n == "ssao_history_ao") p = &chain->ssao->history_ao[ci];
    else if (n == "ssao_history_len") p = &chain->ssao->history_len[ci];
    else if (n == "ae_low_res") p = chain->auto_exposure ? &chain->auto_exposure->low_res : nullptr;
    MIFX_REQUIRE(p != nullptr && p->data != nullptr, "mifx_chain_get_shard_plane: unknown or unallocated plane '%s'", name);
    *out = p->desc();
    return MIFX_OK;
}

mifx_status mifx_chain_set_auto_exposure(mifx_chain* chain, int32_t enable, float elapsed_time_s, int32_t light_adaptation)
{
    MIFX_REQUIRE(chain != nullptr && elapsed_time_s >= 0.0f, "mifx_chain_set_auto_exposure: bad argument");
    if (enable && !chain->auto_exposure) MIFX_CHECK(mifx_autoexposure_create(chain->ctx, &chain->auto_exposure));
    if (!enable && chain->auto_exposure)
    {
        mifx_autoexposure_destroy(chain->auto_exposure);
        chain->auto_exposure = nullptr;
    }
    chain->ae_elapsed = elapsed_time_s;
    chain->ae_adapt   = light_adaptation != 0;
    return MIFX_OK;
}

mifx_status mifx_chain_set_depth_of_field(mifx_chain* chain, const mifx_dof_attribs* attribs, uint32_t feature_flags)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_depth_of_field: null chain");
    if (chain->bloom) chain->bloom->output_deferred = false; // (a deferred Bloom output would read the depth-of-field output plane that is about to go / to change)
    if (attribs == nullptr)
    {
        mifx_dof_destroy(chain->dof);
        chain->dof = nullptr;
        return MIFX_OK;
    }
    MIFX_REQUIRE((feature_flags & ~3u) == 0, "mifx_chain_set_depth_of_field: unknown feature flags 0x%x", feature_flags);
    if (!chain->dof) MIFX_CHECK(mifx_dof_create(chain->ctx, &chain->dof));
    chain->dof_attribs = *attribs;
    chain->dof_flags   = feature_flags;
    return MIFX_OK;
}

mifx_status mifx_chain_set_material_layers(mifx_chain* chain, const mifx_pbr_layers* layers, const mifx_pbr_shadows* shadows)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_material_layers: null chain");
    if (shadows != nullptr)
        MIFX_REQUIRE(shadows->shadow_map != nullptr && shadows->shadow_map_count <= MIFX_PBR_MAX_SHADOW_MAPS && (shadows->shadow_map_count == 0 || shadows->shadow_maps != nullptr),
                     "mifx_chain_set_material_layers: bad shadows block");
    chain->has_layers = chain->has_shadows = false;
    if (layers != nullptr && layers->flags != 0u)
    {
        chain->layers = *layers;
        const mifx_image2d** slot[9] = {&chain->layers.clearcoat, &chain->layers.clearcoat_normal, &chain->layers.sheen, &chain->layers.anisotropy, &chain->layers.tangent,
                                        &chain->layers.iridescence, &chain->layers.transmission, &chain->layers.sheen_albedo_scaling_lut, &chain->layers.preintegrated_charlie};
        for (int i = 0; i < 9; ++i)
            if (*slot[i] != nullptr)
            {
                chain->layer_images[i] = **slot[i];
                *slot[i] = &chain->layer_images[i];
            }
        chain->has_layers = true;
    }
    if (shadows != nullptr)
    {
        chain->shadows      = *shadows;
        chain->shadow_array = *shadows->shadow_map;
        for (uint32_t i = 0; i < shadows->shadow_map_count; ++i) chain->shadow_infos[i] = shadows->shadow_maps[i];
        chain->shadows.shadow_map  = &chain->shadow_array;
        chain->shadows.shadow_maps = chain->shadow_infos;
        chain->has_shadows = true;
    }
    return MIFX_OK;
}

mifx_status mifx_chain_set_effect_feature_flags(mifx_chain* chain, uint32_t ssao_feature_flags, uint32_t ssr_feature_flags)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_effect_feature_flags: null chain");
    MIFX_REQUIRE((ssao_feature_flags & ~3u) == 0 &&
                     (ssr_feature_flags & ~(uint32_t(MIFX_SSR_FEATURE_FLAG_PREVIOUS_FRAME) | uint32_t(MIFX_SSR_FEATURE_FLAG_HALF_RESOLUTION))) == 0,
                 "mifx_chain_set_effect_feature_flags: SSAO 0x%x / SSR 0x%x: unknown flag", ssao_feature_flags, ssr_feature_flags);
    chain->ssao_flags = ssao_feature_flags;
    chain->ssr_flags  = ssr_feature_flags;
    return MIFX_OK;
}

mifx_status mifx_chain_set_postfx_feature_flags(mifx_chain* chain, uint32_t feature_flags)
{
    MIFX_REQUIRE(chain != nullptr && (feature_flags & ~7u) == 0, "mifx_chain_set_postfx_feature_flags: unknown flag 0x%x", feature_flags);
    // (FEATURE_FLAG_TEMPORAL_UPSCALING: the chain has no up-scaler between TAA and Bloom -- neither has HnPostProcessTask -- so Bloom accepts the TAA output only
    //  when FrameDesc.OutputWidth x OutputHeight equals Width x Height; a different output size is refused by mifx_bloom_execute with the sizes in the message)
    chain->postfx_flags = feature_flags;
    return MIFX_OK;
}

mifx_status mifx_chain_get_auto_exposure(mifx_chain* chain, mifx_autoexposure** out)
{
    MIFX_REQUIRE(chain != nullptr && out != nullptr, "mifx_chain_get_auto_exposure: null argument");
    *out = chain->auto_exposure;
    return MIFX_OK;
}

mifx_status mifx_chain_set_fusion(mifx_chain* chain, int32_t tone_map_into_bloom, int32_t ssr_mask_into_shade)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_fusion: null chain");
    chain->fuse_tone_map = tone_map_into_bloom != 0;
    chain->fuse_ssr_mask = ssr_mask_into_shade != 0;
    return MIFX_OK;
}

mifx_status mifx_chain_set_fusion_mask(mifx_chain* chain, uint32_t mask)
{
    MIFX_REQUIRE(chain != nullptr && (mask & ~uint32_t(MIFX_CHAIN_FUSE_EVERY_SWITCH)) == 0, "mifx_chain_set_fusion_mask: bad argument (mask 0x%x)", mask);
    chain->fuse_tone_map    = (mask & MIFX_CHAIN_FUSE_TONE_MAP_INTO_BLOOM) != 0;
    chain->fuse_ssr_mask    = (mask & MIFX_CHAIN_FUSE_SSR_MASK_INTO_SHADE) != 0;
    chain->fuse_ssr_cleanup = (mask & MIFX_CHAIN_FUSE_SSR_CLEANUP_INTO_COMPOSITE) != 0;
    chain->ssao->fused_resolve = (mask & MIFX_CHAIN_FUSE_SSAO_RESOLVE) != 0;
    chain->fuse_bloom_output   = (mask & MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND) != 0;
    chain->fuse_composite_taa  = (mask & MIFX_CHAIN_FUSE_COMPOSITE_INTO_TAA) != 0;
    return MIFX_OK;
}

mifx_status mifx_chain_set_overlap(mifx_chain* chain, int32_t enable)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_overlap: null chain");
    MIFX_REQUIRE(enable >= 0 && enable <= 5, "mifx_chain_set_overlap: %d (0 off, 1 inside a frame, 2 across frames, 3 three lanes across frames, 4 three lanes with two frames in flight, 5 = 4 with the composite and TAA on the Bloom lane)", enable);
    if (chain->overlap != enable) chain->prep_consumed = false; // (a change of the mode: the next frame forks from the context stream once)
    chain->overlap = enable;
    return MIFX_OK;
}

// "waiter<signal@delta,...": see mifx.h.  An empty or NULL list removes every edge.
mifx_status mifx_chain_set_lane_edges(mifx_chain* chain, const char* edges)
{
    MIFX_REQUIRE(chain != nullptr, "mifx_chain_set_lane_edges: null chain");
    std::vector<mifx_chain::Edge> parsed;
    const std::string list = edges ? edges : "";
    size_t pos = 0;
    while (pos < list.size())
    {
        size_t stop = list.find(',', pos);
        if (stop == std::string::npos) stop = list.size();
        const std::string item = list.substr(pos, stop - pos);
        pos = stop + 1;
        if (item.empty()) continue;
        const size_t lt = item.find('<'), at = item.find('@');
        MIFX_REQUIRE(lt != std::string::npos && lt > 0 && at != std::string::npos && at > lt + 1 && at + 1 < item.size(),
                     "mifx_chain_set_lane_edges: '%s' is not of the form waiter<signal@frames", item.c_str());
        const int d = std::atoi(item.c_str() + at + 1);
        MIFX_REQUIRE(d >= 0 && d <= 3, "mifx_chain_set_lane_edges: '%s': 0 .. 3 frames back", item.c_str());
        parsed.push_back(mifx_chain::Edge{item.substr(0, lt), item.substr(lt + 1, at - lt - 1), d});
    }
    chain->edges.swap(parsed);
    chain->prep_consumed = false; // (the next frame forks from the context stream once)
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
