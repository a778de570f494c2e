// mifx_objects.h -- host objects behind the opaque C handles.  Each mirrors one reference class:
//   mifx_postfx == PostFXContext                 (PostProcess/Common/src/PostFXContext.cpp)
//   mifx_ssao   == ScreenSpaceAmbientOcclusion   (PostProcess/ScreenSpaceAmbientOcclusion/src/...cpp)
//   mifx_ssr    == ScreenSpaceReflection         (PostProcess/ScreenSpaceReflection/src/...cpp)
//   mifx_taa    == TemporalAntiAliasing          (PostProcess/TemporalAntiAliasing/src/...cpp)
//   mifx_bloom  == Bloom                         (PostProcess/Bloom/src/Bloom.cpp)
//   mifx_chain  == the canonical caller, HnPostProcessTask (Hydrogent/src/Tasks/HnPostProcessTask.cpp:743-948)
#pragma once
#include "mifx_rows.h"
#include <cmath>
#include <functional>
#include <map>
#include <string>
#include <vector>
#include "mifx_host.h"

struct mifx_postfx
{
    int         device = 0;
    hipStream_t stream = nullptr;

    mifx_frame_desc frame{};
    uint32_t        flags    = 0;
    bool            prepared = false;
    bool            executed = false;

    // blue-noise sampler tables (device copies) and per-frame 128x128 noise planes (C1)
    void*       sobol_dev      = nullptr; // 256 bytes
    void*       scrambling_dev = nullptr; // 128*128*8 bytes
    mifx::Plane noise_xy, noise_zw;
    uint32_t    noise_frame = ~0u;

    // C2/C3 outputs and the C4 alias
    mifx::Plane  reproj_depth, closest_motion;
    mifx_image2d prev_depth{};
    // FEATURE_FLAG_HALF_PRECISION_DEPTH in the native-storage build: the reprojected depth and a copy of the previous depth hold what R16_UNORM targets keep
    // (PostFXContext.cpp:259-270).  Like the reference's formats, decided when the planes are (re)created -- on a change of the frame size, not of the flags (:246-247).
    bool        depth16 = false;
    mifx::Plane prev_depth16;
    mifx_camera_attribs curr_cam{}, prev_cam{};

    // Row-band sharding (mifx_rows.h): `need` = rows of the next effect's final output that its consumers read ({0,0}: the whole frame),
    // `max_motion` = bound on the reprojection reach in rows, `prep_rows` = rows the last mifx_postfx_execute produced.
    mifx::Rows need{0, 0}, prep_rows{0, 0}, band{0, 0}; // band = rows of the final image this rank owns ({0,0}: all)
    int        max_motion = 0;
    mifx::Rows needed_rows(int h) const { return need.empty() ? mifx::Rows{0, h} : mifx::rows_clip(need, h); }

    // Counts the work the library queues on `stream` OUTSIDE an execute call (history fills of a reset or a re-allocating prepare, history imports, table uploads).
    // mifx_chain's multi-stream modes let the lanes of the next frame wait for events of the previous frame only; when this counter moved since the last frame the
    // chain orders every lane behind the context stream once (api_chain.cpp: a full fork).
    uint64_t stream_epoch = 0;
    // Events of work that runs on OTHER streams and touches planes that outside-of-execute work may touch too (the history-halo exchanges of a sharded chain, api_comm.cpp):
    // whatever is queued on `stream` outside an execute call is ordered behind them first.
    std::vector<hipEvent_t> pending_joins;
    void queued_outside_execute()
    {
        ++stream_epoch;
        for (hipEvent_t e : pending_joins) (void)hipStreamWaitEvent(stream, e, 0);
        pending_joins.clear();
    }

    // HIP-event bracket around every launch of one named kernel (mifx_postfx_set_kernel_timing): slot i = i-th launch since it was armed
    std::string             timed_kernel;
    std::vector<hipEvent_t> timed_events; // 2 per slot
    uint32_t                timed_launches = 0;

    // Called at the launch sites of the named kernels (MifxKernelTimer: begin = before the launch, end = behind it), on whatever `stream` is at that moment.  The
    // chain's pipelined mode (api_chain.cpp, mifx_chain_set_overlap 4) records "this kernel of this frame is done" events there and orders a kernel behind a kernel of
    // an earlier frame (MIFX_LANE_EDGES); empty outside that mode.
    std::function<void(const char* name, bool begin)> kernel_hook;

    // per-call working copy of the IBL cube maps with a one-texel apron per face (P6/P7, see pbr.hip); grown on demand
    mifx::IblApronCache ibl_apron;

    ~mifx_postfx();
};

// RAII bracket used at the launch sites of the large kernels: records start / stop events on the launch stream when `name` is the armed kernel
// ... and opens the rocTX range of the reference pass the kernel implements (mifx_host.h MifxRange; names = the reference's ScopedDebugGroup markers)
const char* mifx_reference_pass_name(const char* kernel);
struct MifxKernelTimer
{
    mifx_postfx*    ctx;
    int             slot = -1;
    const char*     hooked = nullptr;
    mifx::MifxRange range;
    MifxKernelTimer(mifx_postfx* c, const char* name) : ctx(c), range(mifx_reference_pass_name(name))
    {
        if (c->kernel_hook)
        {
            hooked = name;
            c->kernel_hook(name, true);
        }
        if (!c->timed_kernel.empty() && c->timed_kernel == name && 2 * (c->timed_launches + 1) <= c->timed_events.size())
        {
            slot = int(c->timed_launches++);
            (void)hipEventRecord(c->timed_events[2 * slot], c->stream);
        }
    }
    void stop()
    {
        if (slot >= 0) (void)hipEventRecord(ctx->timed_events[2 * slot + 1], ctx->stream);
        slot = -1;
        if (hooked && ctx->kernel_hook) ctx->kernel_hook(hooked, false);
        hooked = nullptr;
        range.end();
    }
    ~MifxKernelTimer() { stop(); }
};

struct mifx_ssao
{
    mifx_postfx* ctx = nullptr;
    uint32_t     w = 0, h = 0, flags = 0;
    bool         prepared = false;
    uint32_t     last_frame = ~0u;
    bool         force_reset = true;

    static constexpr int kMips = 5;            // SSAO_DEPTH_PREFILTERED_MAX_MIP + 1
    // Round 6, per-frame requests of a sharded frame (api_comm.cpp; cleared by the execute that takes them).  A2's last level is read anywhere (A3's far taps), so until
    // round 5 every rank reduced the WHOLE depth pyramid -- 46 us of a 1.15 ms band at 7680x4320 / 8 ranks.  With `own_last_level` = the rows of the last level this rank
    // owns (non-empty), A2 reduces only the source rows its own store windows and those rows need, and `after_prefilter` -- the all-gather of the last level's depth and
    // camera-z planes, 1 MB per frame in total at that size -- runs between A2 and A3.  Without the hook (mifx_chain_execute_band: the compute side alone) the rows of
    // other ranks stay stale.
    bool       gather_last_level = false; // the request itself (a thin band may own no row of the last level: own_last_level is then empty)
    mifx::Rows own_last_level{0, 0};
    std::function<mifx_status(const mifx::Plane& depthLast, const mifx::Plane& camzLast, hipStream_t s)> after_prefilter;
    // Row-band sharding: A5's / A6's row window starts and ends on a multiple of this many rows -- one row of A6's last level, what the fused pyramid kernel needs
    // (mifx_pyramid.h first_row(); 32 until round 5, when a launch had to start on one of its own 16-row blocks of level 1)
    static constexpr int kWindowAlign = 1 << (kMips - 1);
    mifx::Plane prefiltered_depth[kMips];      // A2 (mip 0 = copy of the depth)
    mifx::Plane prefiltered_camz[kMips];       // depth_to_camera_z of every level of the depth pyramid (level 0 = of the depth buffer): views into camz_slab
    mifx::DeviceScratch camz_slab;             // one allocation, so that an A3 tap addresses any level with a 32-bit offset from a uniform base
    mifx::Plane checkerboard_depth;            // A1 (FEATURE_FLAG_HALF_RESOLUTION): level 0 of the half-size pyramid
    mifx::Plane full_camz;                     // half resolution only: camera z of the full-size depth for A8 (otherwise prefiltered_camz[0])
    mifx::Plane occlusion_upsampled;           // A4 (half resolution only)
    mifx::Plane occlusion;                     // A3
    mifx::Plane accum_ao;                      // A5 output (the reference writes it into history[curr], which A8's copy then overwrites)
    mifx::Plane history_ao[2], history_len[2]; // ping-pong by FrameDesc.Index & 1: resolved AO (A8) / history length (A5)
    mifx::Plane conv_ao[kMips], conv_depth[kMips]; // A6 (mip 0 aliases are handled in execute)
    mifx::Plane resampled;                     // A7
    // The resolved AO.  Standalone use: `output`, one plane for the life of the object like the reference's OCCLUSION_HISTORY_RESOLVED (GetAmbientOcclusionSRV hands
    // out one resource), written beside history_ao[curr] by the resolve (the reference copies one into the other, .cpp:1319-1328).  Inside mifx_chain nobody keeps the
    // descriptor across frames, so the chain sets alias_output: history_ao[curr] IS the output (one plane and one store per texel less).
    mifx::Plane output;
    bool        alias_output = false;
    // A7 + A8 as one resolve folded into A5 + two work-list passes (ssao.hip: "fused resolve"); off = the two full-frame passes (test hook
    // mifx_debug_ssao_set_fused_resolve, MIFX_SSAO_FUSED_RESOLVE=0)
    bool                fused_resolve = true;
    bool                depth16 = false; // FEATURE_FLAG_HALF_PRECISION_DEPTH in the native-storage build: the two depth pyramids hold what R16_UNORM targets keep
    mifx::DeviceScratch resolve_lists;
};

struct mifx_ssr
{
    mifx_postfx* ctx = nullptr;
    uint32_t     w = 0, h = 0, flags = 0;
    bool         prepared = false;
    uint32_t     last_frame = ~0u;

    static constexpr int kMips = 7; // SSR_DEPTH_HIERARCHY_MAX_MIP + 1
    // Row-band sharding without a radiance exchange (api_chain.cpp, phase 2): R4 records where every ray hit (hit_coords: x | y << 16 per ray texel) instead of
    // loading the colour there, and `after_trace` -- the chain's hit fetch (launch_pbr_hit_fetch) -- runs between R4 and R5.  Both are per-frame requests.
    mifx::Plane hit_coords;
    std::function<mifx_status(mifx::Img rays, mifx::Img coords)> after_trace;
    mifx::Rows  hit_local_rows{0, 0}; // with after_trace: the rows of the colour buffer that are valid on this rank -- R4 loads hits there itself and records only the others
    // R1 on another stream (per-frame request of the chain's lanes mode, mifx_chain_set_overlap 3): the depth hierarchy depends on the depth buffer only; it is
    // recorded on `hiz_stream`, `hiz_done` is recorded behind it and the effect's own stream waits for that event before R2 / R4.
    hipStream_t hiz_stream = nullptr;
    hipEvent_t  hiz_done   = nullptr;
    int         direct_level0 = -1; // the march reads level 0 from the caller's depth plane instead of a copy: -1 = in row-band frames only, 0 = never, 1 = always (api_ssr.cpp)
    mifx::Plane hiz[kMips];         // R1: views into hiz_slab (level 0 = copy of the depth)
    mifx::DeviceScratch hiz_slab;
    mifx::Plane mask_half;          // R3 (FEATURE_FLAG_HALF_RESOLUTION): the mask of the half-size ray pass
    mifx::Plane roughness, mask;    // R2 (mask: 1 float per texel, 1 = reflection sample)
    uint32_t    mask_provided_for = ~0u; // frame index for which the chain's shade kernel has already written `roughness` / `mask` (execute then skips R2)
    // rows of the ray march / of R2 for the rows `need` of the output (derivation in mifx_ssr_execute); the chain's shade covers them when it provides the mask
    // Half resolution: R5's taps of a full-size row y land on the half-size rows int(0.5 (floor(y) +- reach) + 0.5); R4 / R3 run on those (half_rows), and R3 / R4 read
    // the roughness, normal and depth of the 2 x 2 (3 rows at an odd height) full-size block of every such texel: the full-size rows cover them as well.
    static mifx::Rows half_rows(const mifx_ssr_attribs& a, mifx::Rows w5, int H)
    {
        const int reach = int(std::ceil(a.SpatialReconstructionRadius)) + 1;
        return mifx::rows_clip(mifx::Rows{(w5.b - reach) / 2 - 1, (w5.e - 1 + reach) / 2 + 2}, H / 2);
    }
    static mifx::Rows march_rows(const mifx_ssr_attribs& a, mifx::Rows need, int H, bool half = false)
    {
        const mifx::Rows w4 = mifx::rows_expand(need, 3 + 1 + int(std::ceil(a.SpatialReconstructionRadius)) + 1, H);
        if (!half) return w4;
        const mifx::Rows h4 = half_rows(a, mifx::rows_expand(need, 3 + 1, H), H);
        return mifx::rows_hull(w4, mifx::rows_clip(mifx::Rows{2 * h4.b, 2 * h4.e + 1}, H));
    }
    mifx::Plane ray_radiance, ray_dir_pdf;                 // R4
    mifx::Plane res_radiance, res_variance, res_depth;     // R5
    mifx::Plane hist_radiance[2], hist_variance[2];        // R6 ping-pong
    mifx::Plane output;                                    // R7
    // R7 deferred (the chain's composite evaluates the cleanup per pixel instead of reading `output`, mifx_ssr_cleanup.h): mifx_ssr_execute then stops after R6 and
    // keeps what the pass needs; mifx_ssr_get_output runs it on demand (tests, tools), so the plane is still there for whoever asks.
    bool               defer_cleanup   = false; // set by the chain before execute (per frame)
    bool               cleanup_pending = false;
    mifx::SsrCleanupIn cleanup_in{};
    mifx::Img          cleanup_normal{};
    mifx::CamK         cleanup_cam{};
    mifx::Rows         cleanup_rows{0, 0};
    mifx_status        run_cleanup(); // launches R7 if it is pending
};

struct mifx_taa
{
    mifx_postfx* ctx = nullptr;
    uint32_t     w = 0, h = 0, flags = 0;
    bool         prepared = false;
    uint32_t     last_frame = ~0u, curr_frame = 0;
    mifx::Plane  accum[2];
    // The reference decides in PrepareResources whether the technique of the frame's flag set exists and is ready (m_AllPSOsReady, TemporalAntiAliasing.cpp:161-171), and
    // creates it in Execute (:184): the FIRST frame a flag set is executed with takes ComputePlaceholderTexture -- a plain copy of the colour, alpha included, into
    // the accumulation buffer (:191-198, :302-311) -- and counts as history for the next one (UpdateConstantBuffer ran: LastFrameIdx is set).  Found by executing the
    // reference's host code (oracle/refhost, round 4); bit `flags` of techniques_created = that flag set has been executed once.
    uint32_t     techniques_created = 0;
    bool         technique_ready    = false; // of this frame's flag set, as of mifx_taa_prepare
    // Per-frame request of the chain (MIFX_CHAIN_FUSE_COMPOSITE_INTO_TAA): the colour to accumulate is the chain's composite, which the kernel evaluates itself for its
    // colour tile (taa.hip) -- `color` of the render attribs is then not read.  Taken and cleared by mifx_taa_execute; never combined with the placeholder frame (the
    // chain asks technique_ready first).
    const mifx::TaaFusedComposite* fused_composite = nullptr;
};

struct mifx_bloom
{
    mifx_postfx* ctx = nullptr;
    uint32_t     w = 0, h = 0, flags = 0;
    bool         prepared = false;
    std::vector<mifx::Plane*> down, up;
    mifx::Plane  output;
    ~mifx_bloom();
    // Row-band sharding: levels 0 .. gather_level are computed on row windows, down[gather_level] is assembled from all ranks between the two
    // phases (each rank contributes the rows it owns), the coarser levels are tiny and computed whole on every rank.
    static constexpr int kGatherLevel = 1; // (level 2 made the TAA window 34 rows larger than the band on each side, level 1 makes it 13: 4x the gather volume -- 32 MB at 8K -- for 9 % fewer rows in the shade, SSR, composite and TAA of a 420-row band)
    static constexpr uint32_t kTailTexels = 512; // (measured: one workgroup = one CU is slower than the per-level kernels from ~2000 texels up -- 36 us against 30 for the five smallest 4K levels) // levels of at most this many texels are taken down and up again by one workgroup (launch_bloom_tail)
    bool fuse_tail = true;                          // test hook: mifx_debug_bloom_set_tail
    struct Plan
    {
        int        G = -1;          // -1: unsharded (everything whole)
        mifx::Rows taa{0, 0};       // rows of the input colour the prefilter reads
        mifx::Rows own{0, 0};       // rows of down[G] this rank owns
        mifx::Rows down[16], up[16]; // row windows of every level: up to G from this rank's band alone (down[G] = own), beyond G from the gathered level
                                     // (the rows this rank's band needs on the way down to the last level and up again)
        mifx::Rows compute0{0, 0};   // rows of level 0 the prefilter of THIS rank produces: down[0], or -- halo_level0 -- the rows it owns (own0); the others arrive
        mifx::Rows own0{0, 0};       // rows of level 0 whose first full-resolution row lies in the band (a partition of the level over the ranks)
    };
    // Round 6, a request of mifx_chain_execute_sharded (api_comm.cpp) for the duration of a frame.  Bloom's level 0 (half resolution) feeds two consumers beyond a band's
    // own rows: the rows of level 1 the rank owns (+-4 level-0 rows) and the up-sampling of its band (+-3).  Until round 5 every rank produced those rows itself, which
    // made the TAA output -- and with it the shade, SSR, SSAO, the composite -- 13 rows taller than the band on each side.  With halo_level0 a rank prefilters the level-0
    // rows it owns (TAA window: band +- 4) and the ranks exchange the few rows beside the band edges (`after_level0`: 245 KB per neighbour at 7680x4320) between the prefilter
    // and the first down-sampling.  Off for a caller that drives the phases itself (mifx_chain_execute_phase: no hook, no new exchange to know about).
    bool halo_level0 = false;
    std::function<mifx_status(const mifx::Plane& level0, hipStream_t s)> after_level0;
    Plan make_plan(mifx::Rows band, mifx::Rows need, int mipCount) const;
    int  mip_count(const mifx_bloom_attribs& a) const;
    // the chain's copy-frame pass fused into the final up-sample (launch_bloom_final_tonemap): the LDR target and the ToneMap() arguments
    struct FusedToneMap
    {
        const mifx_image2d*              ldr;
        const mifx_tone_mapping_attribs* attribs;
        float                            ave_log_lum;
        uint32_t                         flags;
        bool                             skip_output = false; // write the tone-mapped frame only; `output` is produced on demand (run_deferred_output)
    };
    // MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND: what the plain final up-sample of the last frame needs (the colour plane is borrowed from the caller -- inside the chain
    // the TAA accumulation buffer or the depth-of-field output -- and, like up[0], intact until the next frame)
    bool               output_deferred = false;
    mifx::Img          deferred_color{};
    bool               deferred_packed = false; // (deferred_color is depth of field's R11G11B10 plane: native-storage build)
    mifx_bloom_attribs deferred_attribs{};
    mifx::Rows         deferred_rows{0, 0};
    mifx_status        run_deferred_output();
    mifx_status run(const mifx_bloom_render_attribs* ra, int phase, const FusedToneMap* tone_map = nullptr); // 0: everything, 1: up to the gather, 2: after the gather
};

struct mifx_dof // == DepthOfField (PostProcess/DepthOfField/src/DepthOfField.cpp)
{
    mifx_postfx* ctx = nullptr;
    uint32_t     w = 0, h = 0, flags = 0;
    bool         prepared = false;
    uint32_t     last_pass = 0; // test hook (mifx_debug_dof_set_last_pass)
    mifx::Plane  coc;            // D1: signed CoC
    mifx::Plane  coc_temporal[2]; // D2 ping-pong by FrameDesc.Index & 1 (FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING only)
    mifx::Plane  dilation[3];    // D4 levels 1..3 (level 0 = D3 is folded into the first reduction)
    mifx::Plane  dilation_blurred; // D5 (the reference blurs the last level in place through an intermediate)
    mifx::Plane  prefiltered[2], bokeh[2]; // half resolution, near / far: D6 -> D7 -> D8 (into prefiltered) -> D9 (into bokeh), as in the reference
    mifx::Plane  output;         // D10
    // host tables (DepthOfField.cpp:49-94) and their device copies
    float        gauss[13] = {};
    mifx::DeviceScratch kernel_large, kernel_small; // float2 points
    int          rings = 0, density = 0, large_count = 0, small_count = 0;
    uint32_t     curr_slot = 0;
    // Row-band sharding: the rows of the passes that depend on the colour buffer, from the rows of the output its consumer reads back to the colour rows the effect
    // needs (D1-D5 depend on the depth buffer and the motion vectors only and are computed whole on every rank, history included).  Half-resolution rows unless named:
    //   D10 at row y samples the half-size bokeh planes bilinearly around y / 2 - 0.25            -> half rows [y / 2 - 1, y / 2 + 1]
    //   D9 takes four bilinear taps half a texel around the centre                                  -> +- 1
    //   D8 / D7 sample at uv + (0.25 | 0.5) * kernel * coc * MaxCircleOfConfusion, y scaled by the aspect ratio: up to 0.125 | 0.25 * MaxCoC * width half rows, + 1 bilinear
    //   D6 at half row h reads the colour rows 2 h, 2 h + 1; D7's Karis weights read the colour at the gather taps (inside D6's colour rows)
    // (one row of slack per step for the rounding of the uv arithmetic).
    struct Windows
    {
        mifx::Rows out, h9, h8, h7, h6, colour; // D10 output (full resolution); D9 / D8 / D7 / D6 outputs (half resolution); colour rows read (full resolution)
    };
    static Windows windows(const mifx_dof_attribs& a, mifx::Rows out, int W, int H)
    {
        const int hH = H / 2;
        const int rf = int(std::ceil(0.125 * double(a.MaxCircleOfConfusion) * W)) + 2, rg = int(std::ceil(0.25 * double(a.MaxCircleOfConfusion) * W)) + 2;
        Windows w;
        w.out = mifx::rows_clip(out, H);
        if (w.out.b <= 0 && w.out.e >= H)
        {
            w.h9 = w.h8 = w.h7 = w.h6 = mifx::Rows{0, hH};
            w.colour = mifx::Rows{0, H};
            return w;
        }
        w.h9 = mifx::rows_clip(mifx::Rows{w.out.b / 2 - 2, (w.out.e - 1) / 2 + 3}, hH);
        w.h8 = mifx::rows_expand(w.h9, 2, hH);
        w.h7 = mifx::rows_expand(w.h8, rf, hH);
        w.h6 = mifx::rows_expand(w.h7, rg, hH);
        w.colour = mifx::rows_hull(w.out, mifx::rows_clip(mifx::Rows{2 * w.h6.b, 2 * w.h6.e}, H));
        return w;
    }
};

struct mifx_autoexposure
{
    mifx_postfx* ctx = nullptr;
    mifx::Plane  low_res; // 64x64 F32X2: (LogLum * Weight, Weight), mip 0 of g_tex2DLowResLuminance
    mifx::Plane  average; // 1x1 F32: g_tex2DAverageLuminance
    // Row-band sharding: the low-resolution row sy samples the colour rows around centre_row(sy, H) (one row either side at most), and is written by the rank
    // whose band holds that row; sample_rows() = the low-resolution rows of a band.
    static int        centre_row(int sy, int H) { return int((int64_t(2 * sy + 1) * H) / 128); }
    static mifx::Rows sample_rows(mifx::Rows band, int H)
    {
        int b = 0, e = 0;
        for (int sy = 0; sy < 64; ++sy)
        {
            const int c = centre_row(sy, H);
            b += c < band.b;
            e += c < band.e;
        }
        return mifx::Rows{b, e};
    }
};

struct mifx_chain
{
    mifx_postfx* ctx   = nullptr;
    mifx_ssao*   ssao  = nullptr;
    mifx_ssr*    ssr   = nullptr;
    mifx_taa*    taa   = nullptr;
    mifx_bloom*  bloom = nullptr;
    mifx::Plane  radiance, specular_ibl, composite;
    mifx::Rows   shaded_rows{0, 0}; // rows of `radiance` the last shade of this chain wrote (row-band sharding: what the SSR hit fetch may load instead of shading)
    uint32_t     shaded_frame = ~0u; // ... and the frame index of that shade
    bool         profiling = false, timed = false;
    hipEvent_t   ev[MIFX_CHAIN_STAGE_COUNT + 1] = {};
    // PostFX prep + SSAO do not depend on the shaded radiance: they run on a second stream beside PBR shade + SSR (fork / join with events)
    uint32_t     ssao_flags = 0, ssr_flags = 0; // FEATURE_FLAGS of the two effects (mifx_chain_set_effect_feature_flags; HnPostProcessTaskParams::SSAOFeatureFlags / SSRFeatureFlags)
    uint32_t     postfx_flags = 0;         // PostFXContext::FEATURE_FLAGS of every mifx_postfx_prepare (mifx_chain_set_postfx_feature_flags)
    mifx_dof*    dof = nullptr;            // optional (mifx_chain_set_depth_of_field): between TAA and Bloom, HnPostProcessTask.cpp:899-909
    mifx_dof_attribs dof_attribs{};
    uint32_t     dof_flags = 0;
    // mifx_chain_set_material_layers: deep copies (descriptors included), so that only the device planes are borrowed
    bool                     has_layers = false, has_shadows = false;
    mifx_pbr_layers          layers{};
    mifx_image2d             layer_images[9]{};
    mifx_pbr_shadows         shadows{};
    mifx_shadow_map_array    shadow_array{};
    mifx_pbr_shadow_map_info shadow_infos[MIFX_PBR_MAX_SHADOW_MAPS]{};
    mifx_autoexposure* auto_exposure = nullptr; // optional: fAveLogLum of the final tone map from the average luminance of the Bloom output
    float        ae_elapsed = 0.0f;
    bool         ae_adapt   = true;
    mifx::Rows   band{0, 0};    // row-band sharding: rows of the final image this rank owns ({0,0}: unsharded)
    int          max_motion = 0;
    bool         fuse_tone_map = true; // the copy-frame ToneMap as the tail of Bloom's final up-sample (mifx_chain_set_fusion)
    bool         fuse_ssr_cleanup = true; // R7 (SSR's bilateral cleanup) evaluated inside the composite kernel, its only consumer (mifx_ssr_cleanup.h)
    bool         fuse_bloom_output = true; // the Bloom output plane is not written when the tone map is fused into the final up-sample (produced on demand)
    bool         fuse_composite_taa = false; // the composite (with R7 inside) evaluated by the TAA kernel for its colour tile: the composite plane is neither written nor read (taa.hip).
                                            // OFF by default: measured 85 us SLOWER per 4K frame than the two passes (profiles/r05_ab_composite_into_taa.txt)
    mifx_composite_attribs  pending_composite{};        // ... what chain_composite would have launched, kept for the TAA call of the same frame
    mifx::TaaFusedComposite pending_fused{nullptr, nullptr};
    bool         fuse_ssr_mask = true; // R2 (roughness + reflection mask of SSR) written by the shade kernel, which reads the same material / depth texels
    int          overlap = 0; // opt-in (mifx_chain_set_overlap): 1 = prep + SSAO beside shade + SSR, 2 = and across frames, 3 = three lanes across frames, 4 = three lanes, two frames in flight, 5 = 4 with the composite / TAA / depth of field on the Bloom lane; per-kernel durations then overlap and lose their roofline meaning
    bool         prep_consumed = false; // evPrepConsumed was recorded by the previous frame
    uint64_t     seen_epoch = 0;        // ctx->stream_epoch at the end of the previous frame (a difference = work queued on the context stream in between: full fork)
    hipStream_t  side = nullptr, lane_x = nullptr; // side: prep + SSAO (modes 1, 2), shade + prep + Hi-Z + SSAO (mode 3); lane_x: SSR, composite, TAA, depth of field (mode 3)
    hipEvent_t   evFork = nullptr, evPrep = nullptr, evSsao = nullptr, evPrepConsumed = nullptr, evBloomDone = nullptr, evJoinS = nullptr, evJoinX = nullptr;
    // mifx_chain_set_overlap 4: the three lanes with TWO frames in flight.  Lane S of frame k + 1 (shade, prep, Hi-Z, SSAO) starts when lane X of frame k - 1 has ended,
    // i.e. beside lane X of frame k; what S writes and X reads is double-buffered by trading planes with `shadow` at the start of every frame (the effect objects never
    // notice: kernels of the previous frame hold the old addresses by value).
    struct Shadow
    {
        mifx::Plane radiance, specular_ibl;                                    // the chain's
        mifx::Plane reproj_depth, closest_motion, noise_xy, noise_zw, prev_depth16; // mifx_postfx's
        mifx::Plane roughness, mask, hiz[mifx_ssr::kMips];                     // mifx_ssr's (hiz: views into hiz_slab)
        mifx::DeviceScratch hiz_slab;
    } shadow;
    uint64_t   seq = 0;                 // frames executed in mode 4
    uint32_t   last_index = ~0u;        // FrameDesc.Index of the previous frame (the histories ping-pong by its parity: a frame whose index does not follow stays one deep)
    hipEvent_t evXEnd[2] = {nullptr, nullptr}; // end of lane X of frame seq, by seq & 1 (mode 5: the end of the frame's composite / TAA / depth of field on lane M -- the last readers of what lane S overwrites)
    hipEvent_t evSsrDone = nullptr;            // mode 5: the end of the frame's lane X (SSR R4 .. R6)
    // MIFX_LANE_EDGES="waiter<signal@d,...": the kernel `waiter` of frame k is not started before the kernel `signal` of frame k - d is done (names = MifxKernelTimer's);
    // ordering only, never needed for correctness -- which kernels share the GPU is what the pipelined frame has to choose
    struct Edge
    {
        std::string waiter, signal;
        int         delta;
    };
    std::vector<Edge> edges;
    struct Signal
    {
        hipEvent_t ev[4]  = {nullptr, nullptr, nullptr, nullptr};
        uint64_t   seq[4] = {~0ull, ~0ull, ~0ull, ~0ull};
    };
    std::map<std::string, Signal> signals;
    // mifx_chain_execute_sharded: the communicator (borrowed), the row boundaries of all ranks' bands, fork / join events of the radiance all-gather
    struct mifx_comm* comm = nullptr;
    std::vector<int32_t> cuts;
    // mifx_chain_execute_sharded: the history halos travel on a stream of their own, each as soon as the pass that writes the plane is done (SSAO's after phase 1, SSR's and
    // TAA's after phase 2), and are waited for where the NEXT frame first reads them (phase 1 / phase 2) -- the exchange hides behind the rest of this frame and the start of
    // the next instead of ending the frame.  MIFX_SHARD_ASYNC_HALOS=0: at the end of the frame on the context's stream, as until round 4.
    bool        async_halos = true;
    hipStream_t halo_stream = nullptr;
    hipEvent_t  evAfterP1 = nullptr, evAfterP2 = nullptr, evHaloSsao = nullptr, evHaloRest = nullptr;
    bool        halo_ssao_pending = false, halo_rest_pending = false;
    // mifx_chain_execute_sharded with mifx_chain_set_overlap 3: prep + SSAO (phase 1) run on a lane of their own beside the shade (phase 0) and SSR (the first half of
    // phase 2).  Phase 1 records `sig_after_prep` behind the PostFX prep, phase 2 waits for `wait_before_composite` (the end of SSAO) in front of the composite; both are
    // set for the duration of one call by execute_sharded_impl (api_comm.cpp) and null otherwise.
    hipEvent_t  sig_after_prep = nullptr, wait_before_composite = nullptr;
    // ... and SSR's depth hierarchy, whole-frame streaming work that depends on the depth buffer alone, on a fourth stream beside the shade (mifx_ssr::hiz_stream)
    hipStream_t lane_h = nullptr;
    hipEvent_t  evHiz = nullptr, evJoinH = nullptr;
    void        join_halos(); // the context's stream waits for both exchanges (before anything that is not a frame of this chain touches the history planes)
    ~mifx_chain();
};

namespace mifx
{
// what a rank owning the rows `band` of the frame has to receive between the phases (api_chain.cpp); mifx_chain_get_shard_info = this for the chain's own band
mifx_shard_info chain_shard_info(const mifx_chain* chain, const mifx_chain_frame* f, Rows band);
mifx_bloom::Plan chain_bloom_plan(const mifx_chain* chain, const mifx_chain_frame* f, Rows band);
bool            shard_bloom_halo_enabled();
// HnPostProcessTask::Prepare: the per-frame PrepareResources of every effect and the chain's own planes (idempotent for an unchanged frame description)
mifx_status chain_prepare_resources(mifx_chain* chain, const mifx_chain_frame* f);
// the chain's extra streams and their events, created on first use; whether this frame's lanes may start behind the previous frame's events alone (api_chain.cpp)
mifx_status chain_make_lanes(mifx_chain* chain, bool three);
bool        chain_lanes_continue(mifx_chain* chain);
// the depth hierarchy of a W x H frame as one allocation + per-level views (api_ssr.cpp)
mifx_status ssr_alloc_hiz(uint32_t W, uint32_t H, Plane* hiz, DeviceScratch& slab);
// the chain stops borrowing its communicator (api_comm.cpp)
void chain_detach_comm(mifx_chain* chain);
} // namespace mifx
