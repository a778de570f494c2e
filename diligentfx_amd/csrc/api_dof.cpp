// api_dof.cpp -- C ABI + host sequencing of the depth-of-field effect (PostProcess/DepthOfField/src/DepthOfField.cpp), SURVEY 8f N1.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "mifx_objects.h"

using namespace mifx;

namespace
{
constexpr int kKernelCapacity = 128; // width of the reference's large kernel texture (DepthOfField.cpp:110)

// GenerateKernelPoints (DepthOfField.cpp:49-74), "Octaweb": rings from the outside in, the centre point last
std::vector<float> generate_kernel_points(int rings, int density)
{
    std::vector<float> k;
    const float radiusInc = 1.0f / (float(rings) - 1.0f);
    for (int i = rings - 1; i >= 0; --i)
    {
        const int   points   = std::max(density * i, 1);
        const float radius   = float(i) * radiusInc;
        const float thetaInc = 2.0f * 3.14159265358979323846f / float(points);
        const float offset   = 0.1f * float(i);
        for (int j = 0; j < points; ++j)
        {
            const float theta = offset + float(j) * thetaInc;
            k.push_back(radius * std::cos(theta));
            k.push_back(radius * std::sin(theta));
        }
    }
    return k;
}
int kernel_sample_count(int rings, int density) { return 1 + density * ((rings - 1) * rings >> 1); } // ComputeSampleCount, DOF_Common.fx:4-7

mifx_status upload(mifx_postfx* ctx, DeviceScratch& dst, const std::vector<float>& src)
{
    MIFX_CHECK(dst.reserve(src.size() * sizeof(float)));
    ctx->queued_outside_execute();
    // pageable source: the copy is staged before hipMemcpyAsync returns, so the vector may go out of scope
    MIFX_HIP_CHECK(hipMemcpyAsync(dst.data, src.data(), src.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    return MIFX_OK;
}
} // namespace

extern "C" {

mifx_status mifx_dof_generate_kernel_points(int32_t ring_count, int32_t ring_density, float* out, uint32_t capacity_points, uint32_t* out_count)
{
    MIFX_REQUIRE(out != nullptr && out_count != nullptr, "mifx_dof_generate_kernel_points: null argument");
    MIFX_REQUIRE(ring_count >= 2 && ring_density >= 1, "mifx_dof_generate_kernel_points: %d rings x density %d (at least 2 rings: the ring spacing is 1 / (rings - 1))", ring_count,
                 ring_density);
    const int count = kernel_sample_count(ring_count, ring_density);
    MIFX_REQUIRE(count <= kKernelCapacity && uint32_t(count) <= capacity_points, "mifx_dof_generate_kernel_points: %d points exceed the capacity (%u; the reference's texture holds %d)",
                 count, capacity_points, kKernelCapacity);
    const std::vector<float> k = generate_kernel_points(ring_count, ring_density);
    std::memcpy(out, k.data(), k.size() * sizeof(float));
    *out_count = uint32_t(count);
    return MIFX_OK;
}

mifx_status mifx_dof_create(mifx_postfx* ctx, mifx_dof** out)
{
    MIFX_REQUIRE(ctx != nullptr && out != nullptr, "mifx_dof_create: null argument");
    *out = nullptr;
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    mifx_dof* fx = new mifx_dof();
    fx->ctx = ctx;
    { // GenerateGaussKernel(DOF_GAUSS_KERNEL_RADIUS, DOF_GAUSS_KERNEL_SIGMA), DepthOfField.cpp:76-94, 152-170
        const int   radius = 6;
        const float sigma  = 5.0f;
        float       sum    = 0.0f;
        for (int i = -radius; i <= radius; ++i)
        {
            fx->gauss[i + radius] = std::exp(-float(i * i) / (2.0f * sigma * sigma));
            sum += fx->gauss[i + radius];
        }
        for (float& v : fx->gauss) v /= sum;
    }
    fx->small_count = kernel_sample_count(3, 5); // DOF_BOKEH_KERNEL_SMALL_RING_COUNT / _DENSITY (DepthOfField.cpp:131-150)
    const mifx_status st = upload(ctx, fx->kernel_small, generate_kernel_points(3, 5));
    if (st < 0)
    {
        delete fx;
        return st;
    }
    *out = fx;
    return MIFX_OK;
}
void mifx_dof_destroy(mifx_dof* fx) { delete fx; }

mifx_status mifx_debug_dof_set_last_pass(mifx_dof* fx, uint32_t last_pass)
{
    MIFX_REQUIRE(fx != nullptr && last_pass <= 10u, "mifx_debug_dof_set_last_pass: bad argument");
    fx->last_pass = last_pass;
    return MIFX_OK;
}

// DepthOfField::PrepareResources (DepthOfField.cpp:175-293)
mifx_status mifx_dof_prepare(mifx_dof* fx, mifx_postfx* ctx, uint32_t feature_flags)
{
    MIFX_REQUIRE(fx != nullptr && ctx != nullptr, "mifx_dof_prepare: null argument");
    if (!ctx->prepared)
    {
        set_error("mifx_dof_prepare: mifx_postfx_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    MIFX_REQUIRE((feature_flags & ~3u) == 0, "mifx_dof_prepare: unknown feature flags 0x%x", feature_flags);
    fx->ctx       = ctx;
    fx->curr_slot = ctx->frame.Index & 1u; // m_CurrentFrameIdx (:180)
    const uint32_t W = ctx->frame.Width, H = ctx->frame.Height;
    MIFX_REQUIRE(W >= 16 && H >= 16, "mifx_dof_prepare: frame %ux%u too small for the three dilation levels", W, H);
    if (fx->prepared && fx->w == W && fx->h == H && fx->flags == feature_flags) return MIFX_OK;
    fx->prepared = false; // ready again only when every plane of the new size exists (see mifx_ssao_prepare)
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    ctx->queued_outside_execute();
    MIFX_CHECK(fx->coc.alloc(W, H, MIFX_PLANE_COC));
    for (Plane& p : fx->coc_temporal)
    {
        if (feature_flags & MIFX_DOF_FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING)
        {
            MIFX_CHECK(p.alloc(W, H, MIFX_PLANE_COC));
            MIFX_CHECK(p.fill(ctx->stream, 0.0f)); // cleared when (re)created (:205-223)
        }
        else
            p.release();
    }
    for (uint32_t k = 1; k <= 3; ++k) MIFX_CHECK(fx->dilation[k - 1].alloc(W >> k, H >> k, MIFX_PLANE_COC_DILATION));
    MIFX_CHECK(fx->dilation_blurred.alloc(W >> 3, H >> 3, MIFX_PLANE_COC_DILATION));
    for (Plane& p : fx->prefiltered) MIFX_CHECK(p.alloc(W / 2u, H / 2u, MIFX_FORMAT_F32X4));
    for (Plane& p : fx->bokeh) MIFX_CHECK(p.alloc(W / 2u, H / 2u, MIFX_FORMAT_F32X4));
    MIFX_CHECK(fx->output.alloc(W, H, MIFX_PLANE_BLOOM)); // (native-storage build: R11G11B10_FLOAT, DepthOfField.cpp:281-289 -- Bloom reads it as such; fp32 build: F32X4)
    fx->w = W; fx->h = H; fx->flags = feature_flags;
    fx->prepared = true;
    return MIFX_OK;
}

// DepthOfField::Execute (DepthOfField.cpp:295-332; pass bindings :820-1114)
mifx_status mifx_dof_reset_history(mifx_dof* fx)
{
    MIFX_REQUIRE(fx != nullptr, "mifx_dof_reset_history: null argument");
    if (!fx->prepared || fx->ctx == nullptr) return MIFX_OK;
    MIFX_HIP_CHECK(hipSetDevice(fx->ctx->device));
    fx->ctx->queued_outside_execute(); // (fills on the context's stream between two frames: the lanes of the next frame are ordered behind them)
    for (Plane& p : fx->coc_temporal)
        if (p.data) MIFX_CHECK(p.fill(fx->ctx->stream, 0.0f)); // as when the targets are created (DepthOfField.cpp:205-223)
    return MIFX_OK;
}

mifx_status mifx_dof_execute(mifx_dof* fx, const mifx_dof_render_attribs* ra)
{
    MIFX_REQUIRE(fx != nullptr && ra != nullptr && ra->attribs != nullptr && ra->color != nullptr && ra->depth != nullptr, "mifx_dof_execute: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_dof_execute: mifx_dof_prepare must be called first");
        return MIFX_ERR_INVALID_OP;
    }
    mifx_postfx* ctx = fx->ctx;
    MIFX_REQUIRE(ra->postfx == nullptr || ra->postfx == ctx, "mifx_dof_execute: a different PostFX context than the one the resources were prepared with");
    const mifx_dof_attribs& a = *ra->attribs;
    MIFX_REQUIRE(a.BokehKernelRingCount >= 2 && a.BokehKernelRingDensity >= 1, "mifx_dof_execute: %d rings x density %d", a.BokehKernelRingCount, a.BokehKernelRingDensity);
    const int count = kernel_sample_count(a.BokehKernelRingCount, a.BokehKernelRingDensity);
    MIFX_REQUIRE(count <= kKernelCapacity, "mifx_dof_execute: %d kernel points; the reference's kernel texture holds %d (DepthOfField.cpp:110)", count, kKernelCapacity);
    MIFX_REQUIRE(a.MaxCircleOfConfusion > 0.0f, "mifx_dof_execute: MaxCircleOfConfusion must be positive");
    const bool temporal = (fx->flags & MIFX_DOF_FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING) != 0;
    if (!ctx->executed)
    {
        set_error("mifx_dof_execute: the camera (and, for temporal smoothing, the closest motion vectors) come from the PostFX context; mifx_postfx_execute must run first");
        return MIFX_ERR_INVALID_OP;
    }
    Img color, depth;
    MIFX_CHECK(to_img_wh(ra->color, MIFX_FORMAT_F32X4, fx->w, fx->h, "color", color));
    MIFX_CHECK(to_img_wh(ra->depth, MIFX_FORMAT_F32, fx->w, fx->h, "depth", depth));
    MIFX_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    // UpdateConstantBuffers (:799-809): a new (ring count, density) regenerates the large kernel
    if (a.BokehKernelRingCount != fx->rings || a.BokehKernelRingDensity != fx->density)
    {
        MIFX_CHECK(upload(ctx, fx->kernel_large, generate_kernel_points(a.BokehKernelRingCount, a.BokehKernelRingDensity)));
        fx->rings = a.BokehKernelRingCount; fx->density = a.BokehKernelRingDensity; fx->large_count = count;
    }
    const mifx_camera_attribs& cam = ctx->curr_cam;
    const uint32_t last = fx->last_pass ? fx->last_pass : 10u;
    // row windows of the colour-dependent passes (mifx_dof::windows; whole frame by default); D1-D5 are always whole
    const mifx_dof::Windows rw = mifx_dof::windows(a, ctx->needed_rows(int(fx->h)), int(fx->w), int(fx->h));
    // D1
    MIFX_CHECK(launch_dof_coc(s, depth, fx->coc.view(), cam, a.MaxCircleOfConfusion));
    if (last == 1u) return MIFX_OK;
    // D2 (:848-878): history slot = FrameDesc.Index & 1
    const Plane* used = &fx->coc;
    if (temporal)
    {
        const uint32_t cur = fx->curr_slot, prv = cur ^ 1u;
        MIFX_CHECK(launch_dof_temporal_coc(s, fx->coc.view(), fx->coc_temporal[prv].view(), ctx->closest_motion.view(), fx->coc_temporal[cur].view(), cam, a.TemporalStabilityFactor));
        used = &fx->coc_temporal[cur];
    }
    if (last == 2u) return MIFX_OK;
    // D3 + D4 (:879-925)
    const Img levels[3] = {fx->dilation[0].view(), fx->dilation[1].view(), fx->dilation[2].view()};
    MIFX_CHECK(launch_dof_dilation(s, used->view(), levels));
    if (last <= 4u) return MIFX_OK;
    // D5 (:926-972)
    MIFX_CHECK(launch_dof_blur(s, levels[2], fx->dilation_blurred.view(), fx->gauss));
    if (last == 5u) return MIFX_OK;
    // D6 (:973-1006)
    MIFX_CHECK(launch_dof_prefilter(s, color, used->view(), fx->dilation_blurred.view(), win(fx->prefiltered[0].view(), rw.h6), win(fx->prefiltered[1].view(), rw.h6)));
    if (last == 6u) return MIFX_OK;
    // D7 (:1007-1034)
    const float aspect = cam.f4ViewportSize[0] * cam.f4ViewportSize[3];
    {
        MifxKernelTimer timer(ctx, "dof_bokeh_gather_kernel");
        MIFX_CHECK(launch_dof_bokeh_gather(s, fx->prefiltered[0].view(), fx->prefiltered[1].view(), color, win(fx->bokeh[0].view(), rw.h7), win(fx->bokeh[1].view(), rw.h7),
                                           static_cast<const float*>(fx->kernel_large.data), fx->large_count, a.MaxCircleOfConfusion, aspect,
                                           (fx->flags & MIFX_DOF_FEATURE_FLAG_ENABLE_KARIS_INVERSE) != 0));
    }
    if (last == 7u) return MIFX_OK;
    // D8 (:1035-1061): back into the prefiltered textures
    MIFX_CHECK(launch_dof_bokeh_fill(s, fx->bokeh[0].view(), fx->bokeh[1].view(), win(fx->prefiltered[0].view(), rw.h8), win(fx->prefiltered[1].view(), rw.h8), static_cast<const float*>(fx->kernel_small.data),
                                     fx->small_count, a.MaxCircleOfConfusion, aspect));
    if (last == 8u) return MIFX_OK;
    // D9 (:1062-1083): back into the bokeh textures
    MIFX_CHECK(launch_dof_postfilter(s, fx->prefiltered[0].view(), fx->prefiltered[1].view(), win(fx->bokeh[0].view(), rw.h9), win(fx->bokeh[1].view(), rw.h9)));
    if (last == 9u) return MIFX_OK;
    // D10 (:1084-1114)
    return launch_dof_combine(s, color, fx->bokeh[0].view(), fx->bokeh[1].view(), win(fx->output.view(), rw.out), a.AlphaInterpolation);
}

mifx_status mifx_dof_get_output(mifx_dof* fx, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && out != nullptr, "mifx_dof_get_output: null argument");
    if (!fx->prepared)
    {
        set_error("mifx_dof_get_output: resources are not prepared");
        return MIFX_ERR_INVALID_OP;
    }
    *out = fx->output.desc();
    return MIFX_OK;
}

mifx_status mifx_dof_get_intermediate(mifx_dof* fx, const char* name, mifx_image2d* out)
{
    MIFX_REQUIRE(fx != nullptr && name != nullptr && out != nullptr, "mifx_dof_get_intermediate: null argument");
    const std::string n = name;
    const Plane* p = nullptr;
    if (n == "coc") p = &fx->coc;
    else if (n == "coc_temporal") p = &fx->coc_temporal[fx->curr_slot];
    else if (n == "dilation1") p = &fx->dilation[0];
    else if (n == "dilation2") p = &fx->dilation[1];
    else if (n == "dilation3") p = &fx->dilation[2];
    else if (n == "dilation_blurred") p = &fx->dilation_blurred;
    else if (n == "prefiltered0") p = &fx->prefiltered[0];
    else if (n == "prefiltered1") p = &fx->prefiltered[1];
    else if (n == "bokeh0") p = &fx->bokeh[0];
    else if (n == "bokeh1") p = &fx->bokeh[1];
    MIFX_REQUIRE(p != nullptr && p->data != nullptr, "mifx_dof_get_intermediate: unknown or unallocated plane '%s'", name);
    *out = p->desc();
    return MIFX_OK;
}

} // extern "C"
