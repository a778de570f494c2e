// ssr.hip -- ScreenSpaceReflection passes R1..R7 (AMD-SSSR-derived stochastic screen-space reflections).
// Math follows Shaders/PostProcess/ScreenSpaceReflection/private/SSR_*.fx; host sequence in api_ssr.cpp.
//
// Masking: the reference marks reflection samples in a D16 depth target and depth-tests R4-R7 against it
// (ScreenSpaceReflection.cpp:47-51,550,626,...).  Here the mask is a float plane (1 = reflection sample) and every masked pass writes 0
// to masked-out texels of the targets the reference clears every frame (R4's two, R7's output); R5's and R6's targets keep their previous content there, as in the
// reference, whose depth test skips those fragments (since late round 4; see ssr_spatial_kernel).
#include "mifx_host.h"
#include "mifx_effects.h"
#include "mifx_pyramid.h"
#include "mifx_pbr.h"

namespace mifx
{


// ------------------------------------------------------------------------------------------------ R1: Hi-Z mip (SSR_ComputeHierarchicalDepthBuffer.fx:24-71)
MIFX_D float hiz_mip_value(const Img& src, int x, int y, int reversed) // the texel (x, y) of the level below `src`
{
    const int  rx = 2 * x, ry = 2 * y;
    const bool oddW = (src.w & 1) != 0, oddH = (src.h & 1) != 0;
    float m = reversed ? 0.0f : 1.0f; // DepthFarPlane
    auto  tap = [&](int ox, int oy) { m = closest_depth(m, ld_clamp<float>(src, rx + ox, ry + oy), reversed != 0); };
    tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);
    return m;
}
MIFX_D void hiz_mip_texel(const Img& src, const Img& dst, int x, int y, int reversed) { st<float>(dst, x, y, hiz_mip_value(src, x, y, reversed)); }
// Round 6: the LAST TWO levels of the hierarchy in one launch, with no dependency between them.  The closest depth is a minimum (a maximum with reversed depth) that starts
// from the far plane: exact and order-independent, so a texel of level k + 2 is the same reduction over the union of the level-k footprints of its level-(k + 1) taps --
// with the clamps of both steps, evaluated as the reference's two passes would (the taps of the middle level are recomputed, up to nine of up to nine loads: the two
// levels hold 10 000 texels at 3840x2160).  One workgroup per 256 texels of either level; the one-workgroup tail this replaces for whole images took 14.6 us of the lane
// that feeds the ray march, the two single-level launches of a row band 2 x 4.8 us.
__global__ __launch_bounds__(256) void ssr_hiz_last_two_kernel(Img src, Img mid, Img last, int midBlocks, int reversed)
{
    const int b = int(blockIdx.x);
    if (b < midBlocks)
    {
        const int i = b * 256 + int(threadIdx.x);
        if (i < mid.w * mid.h) hiz_mip_texel(src, mid, i % mid.w, i / mid.w, reversed);
        return;
    }
    const int i = (b - midBlocks) * 256 + int(threadIdx.x);
    if (i >= last.w * last.h) return;
    const int  x = i % last.w, y = i / last.w, rx = 2 * x, ry = 2 * y;
    const bool oddW = (mid.w & 1) != 0, oddH = (mid.h & 1) != 0;
    float m = reversed ? 0.0f : 1.0f;
    auto  tap = [&](int ox, int oy) { m = closest_depth(m, hiz_mip_value(src, clampi(rx + ox, 0, mid.w - 1), clampi(ry + oy, 0, mid.h - 1), reversed), reversed != 0); };
    tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);
    st<float>(last, x, y, m);
}
__global__ __launch_bounds__(256) void ssr_hiz_mip_kernel(Img src, Img dst, int reversed)
{
    int x, y;
    if (!pixel_xy(dst, x, y)) return;
    hiz_mip_texel(src, dst, x, y, reversed);
}
// The last levels of the hierarchy, whose sources have odd sizes (3-wide taps: the fused 2x2 kernel does not take them), in ONE workgroup instead of one ~5 us launch
// each: level after level with a workgroup barrier in between (a level is read back by the workgroup that wrote it: stores are complete at the barrier, and the
// texels were never in this CU's L1 before).  Same per-texel code as ssr_hiz_mip_kernel.  At 3840x2160: levels 5 and 6 (120x67, 60x33).
struct HizTail
{
    Img lv[8]; // lv[0] = the source of the first tail level
    int count; // tail levels 1 .. count - 1 are produced
};
__global__ __launch_bounds__(1024) void ssr_hiz_tail_kernel(HizTail t, int reversed)
{
    for (int l = 1; l < t.count; ++l)
    {
        const Img src = t.lv[l - 1], dst = t.lv[l];
        for (int i = int(threadIdx.x); i < dst.w * dst.h; i += int(blockDim.x)) hiz_mip_texel(src, dst, i % dst.w, i / dst.w, reversed);
        __threadfence_block();
        __syncthreads();
    }
}

struct HizOp
{
    using T = float;
    int reversed;           // SSR_OPTION_INVERTED_DEPTH: closest = largest depth, far plane = 0
    Img src, dst[4], copy0; // copy0.p != null: the source level is also written out (level 0 of the hierarchy = a copy of the depth buffer)
    int pairs;              // src and copy0 allow 8-byte accesses (pair_aligned)
    MIFX_D float load(int x, int y) const
    {
        const float v = ld<float>(src, x, y);
        if (copy0.p) st<float>(copy0, x, y, v); // every source texel is read by exactly one thread (even dimensions)
        return v;
    }
    MIFX_D void quad(int x, int y, float& a, float& b, float& c, float& d) const
    {
        if (pairs)
        {
            const v2 r0 = ld_pair(src, 2 * x, 2 * y), r1 = ld_pair(src, 2 * x, 2 * y + 1);
            if (copy0.p) { st_pair(copy0, 2 * x, 2 * y, r0); st_pair(copy0, 2 * x, 2 * y + 1, r1); }
            a = r0.x; b = r1.x; c = r0.y; d = r1.y;
        }
        else { a = load(2 * x, 2 * y); b = load(2 * x, 2 * y + 1); c = load(2 * x + 1, 2 * y); d = load(2 * x + 1, 2 * y + 1); }
    }
    MIFX_D float reduce(float a, float b, float c, float d) const // DepthFarPlane = 1 (0 when reversed)
    {
        return reversed ? fmaxf(fmaxf(fmaxf(fmaxf(0.0f, a), b), c), d) : fminf(fminf(fminf(fminf(1.0f, a), b), c), d);
    }
    MIFX_D float stored(float v) const { return v; }
    MIFX_D bool  inside(int l, int x, int y) const { return x < dst[l - 1].w && y < row_end(dst[l - 1]); }
    MIFX_D int   first_row() const { return dst[0].y0; }
    MIFX_D void  store(int l, int x, int y, float v) const { st<float>(dst[l - 1], x, y, v); }
};
__global__ __launch_bounds__(256) void ssr_hiz_levels_kernel(HizOp op, int nl) { pyramid_reduce_levels(op, nl); }

// ------------------------------------------------------------------------------------------------ R2: mask + roughness (SSR_ComputeStencilMaskAndExtractRoughness.fx:13-40)
__global__ __launch_bounds__(256) void ssr_mask_roughness_kernel(Img material, Img depthTex, Img roughnessOut, Img maskOut, SsrK k)
{
    int x, y;
    if (!pixel_xy(maskOut, x, y)) return;
    const v4 m = ld<v4>(material, x, y);
    const v4 sel{k.RoughnessChannel == 0u ? 1.0f : 0.0f, k.RoughnessChannel == 1u ? 1.0f : 0.0f, k.RoughnessChannel == 2u ? 1.0f : 0.0f, k.RoughnessChannel == 3u ? 1.0f : 0.0f};
    float r = dot(m, sel);
    if (!k.IsRoughnessPerceptual) r = fsqrt(r);
    const float d = ld<float>(depthTex, x, y);
    st<rough_t>(roughnessOut, x, y, r); // every texel (the reference leaves non-sample texels stale)
    st<mask_t>(maskOut, x, y, is_reflection_sample(r, d, k.RoughnessThreshold, k.ReversedDepth != 0) ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------------------------------------ R3: half-resolution mask (SSR_ComputeDownsampledStencilMask.fx:13-61)
__global__ __launch_bounds__(256) void ssr_downsampled_mask_kernel(Img roughnessTex, Img depthTex, Img maskOut, SsrK k)
{
    int x, y;
    if (!pixel_xy(maskOut, x, y)) return;
    const bool rev = k.ReversedDepth != 0, oddW = (depthTex.w & 1) != 0, oddH = (depthTex.h & 1) != 0;
    float minDepth = rev ? 0.0f : 1.0f, maxRough = 0.0f; // DepthFarPlane
    auto tap = [&](int ox, int oy) {
        const int lx = clampi(2 * x + ox, 0, depthTex.w - 1), ly = clampi(2 * y + oy, 0, depthTex.h - 1); // ClampScreenCoord
        minDepth = closest_depth(minDepth, ld<float>(depthTex, lx, ly), rev);
        maxRough = fmaxf(maxRough, ld<rough_t>(roughnessTex, lx, ly));
    };
    tap(0, 0); tap(1, 0); tap(0, 1); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);
    st<mask_t>(maskOut, x, y, is_reflection_sample(maxRough, minDepth, k.RoughnessThreshold, rev) ? 1.0f : 0.0f);
}

// ------------------------------------------------------------------------------------------------ R5: spatial reconstruction (SSR_ComputeSpatialReconstruction.fx:60-175)
static constexpr float c_ssr_poisson[8][3] = {{-0.4706069f, -0.4427112f, +0.6461146f}, {-0.9057375f, +0.3003471f, +0.9542373f}, {-0.3487388f, +0.4037880f, +0.5335386f},
                                          {+0.1023042f, +0.6439373f, +0.6520134f}, {+0.5699277f, +0.3513750f, +0.6695386f}, {+0.2939128f, -0.1131226f, +0.3149309f},
                                          {+0.7836658f, -0.4208784f, +0.8895339f}, {+0.1564120f, -0.8198990f, +0.8346850f}};

// HALF = SSR_OPTION_HALF_RESOLUTION: the Poisson taps address the half-size ray textures (:153-154)
// (5 waves per SIMD: the allocator otherwise keeps all sixteen tap texels and the BRDF frame in 108 registers = 4 waves.  Measured over 60 frames at the steady-state
//  clock in round 3: no hint 155.5 us, 5 waves 142.9 us, 6 waves 207.8 us (spills); profiles/r03_ab_occupancy_hints.txt.  Round 2's "4 / 5: neutral" came from 10-frame runs.)
// (Round 4, measured and not taken: the sixteen tap texels out of an LDS window -- 32 x 8 pixels per workgroup, the 40 x 16 texels of the two ray planes copied in with
//  row-contiguous loads, taps as ds_read_b128; bit-identical, every SSR / sharding / chain test green.  The idea came from tools/microbench/tcp_gather_rate.hip: a CU's
//  vector L1 serves one tag look-up per clock, a wave64 load whose lanes touch >= 16 lines holds it for 64 clocks, and this pass' 57 M look-ups per launch are 93 of its
//  144 us.  Result: 180.3 us against 144.2 us (profiles/r04_ab_r5_window.txt) -- the fill, its barrier and five resident workgroups per CU cost more than the look-ups
//  they replace; the L1 is busy 65 % of this pass, not its limit.)
#ifndef MIFX_R5_WAVES
#define MIFX_R5_WAVES 5
#endif
template <bool HALF> __global__ __launch_bounds__(256) MIFX_WAVES_OPT(MIFX_R5_WAVES) void ssr_spatial_kernel(Img roughnessTex, Img normalTex, Img depthTex, Img dirPdfTex, Img specTex, Img mask, Img outRad, Img outVar,
                                                          Img outDepth, CamK cam, SsrK k)
{
    int x, y;
    if (!pixel_xy_dir<1>(outRad, x, y)) return;
    // Outside the reflection mask the reference's depth test rejects the fragment and the three targets keep what an earlier frame wrote there (they are never cleared,
    // ScreenSpaceReflection.cpp:904-932) -- and R6's statistics and R7's taps READ such texels beside the mask's edge.  Rounds 1-3 wrote 0 here ("stale = undefined"); the executed
    // reference (oracle/refhost, random sequences) showed 0.1-0.2 % of the SSR output depending on it, so the texel is left alone like there.  The planes are zero when created.
    if (ld<mask_t>(mask, x, y) == 0.0f) return;
    const int W = int(cam.vw), H = int(cam.vh);
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    // Memory-level parallelism (round 3): the pass is a chain of dependent round trips -- mask, then the pixel's own depth / normal / roughness, then eight taps whose
    // positions follow from the roughness -- and the counters show its waves parked on s_waitcnt for 62 % of their cycles (profiles/r03_pmc_sq_*: 20 % of the VALU
    // issue roof).  The eight taps used to be eight round trips of their own: each tap's loads were issued, waited for and consumed behind a branch before the next
    // tap's addresses existed.  Now the roughness comes first, the tap positions follow from it alone, and the ray / colour texels of MIFX_R5_BATCH taps are in flight
    // together (and beside the pixel's depth and normal) before anything is consumed; the taps are then accumulated in the reference's order with the same
    // arithmetic -- the "no ray" case is a select instead of a branch -- so every value is what it was.
    const float rough = ld<rough_t>(roughnessTex, x, y);
    const float depth = ld<float>(depthTex, x, y);
    const v3 N        = xyz(ld<v4>(normalTex, x, y));
    const float radius = lerpf(0.0f, k.SpatialReconstructionRadius, saturate(5.0f * rough)); // SSR_SPATIAL_RECONSTRUCTION_ROUGHNESS_FACTOR
    const float angle = 2.0f * MIFX_PI * bayer4x4(unsigned(x), unsigned(y), cam.frameIndex);
    // note: ComputeBlurKernelRotation uses M_PI (3.14159265358979) -- same fp32 value as MIFX_PI
    float sinA, cosA;
    m_sincos(angle, sinA, cosA); // angle in [0, 2 pi)
    const v4 rot{cosA, sinA, -sinA, cosA};
    int tapX[8], tapY[8];
#pragma unroll
    for (int s = 0; s < 8; ++s)
    {
        const v2 xi = rotate_vector(rot, v2{c_ssr_poisson[s][0], c_ssr_poisson[s][1]});
        tapX[s] = HALF ? clampi(int(0.5f * (floorf(pos.x) + radius * xi.x) + 0.5f), 0, int(0.5f * cam.vw) - 1) : clampi(int(pos.x + radius * xi.x), 0, W - 1);
        tapY[s] = HALF ? clampi(int(0.5f * (floorf(pos.y) + radius * xi.y) + 0.5f), 0, int(0.5f * cam.vh) - 1) : clampi(int(pos.y + radius * xi.y), 0, H - 1);
    }
#ifndef MIFX_R5_BATCH
#define MIFX_R5_BATCH 8
#endif
    v4 rayTexel[MIFX_R5_BATCH], colTexel[MIFX_R5_BATCH];
#pragma unroll
    for (int s = 0; s < MIFX_R5_BATCH; ++s) rayTexel[s] = ld<v4>(dirPdfTex, tapX[s], tapY[s]);
#pragma unroll
    for (int s = 0; s < MIFX_R5_BATCH; ++s) colTexel[s] = ld<v4>(specTex, tapX[s], tapY[s]);

    const v3 camPos{cam.pos[0], cam.pos[1], cam.pos[2]};
    const v3 posWS  = inv_project_position(v3{pos.x * cam.ivw, pos.y * cam.ivh, depth}, cam.viewProjInv);
    const v3 V      = normalize(camPos - posWS);
    const float NdotV = saturate(dot(N, V));
    const float alpha = rough * rough;
    const float visV  = smith_ggx_visibility_v_term(NdotV, alpha); // per-pixel factor of the visibility term, hoisted out of the 8-sample loop
    v4    colorSum = mk4(0.0f);
    float weightSum = 0.0f, variance = 0.0f, mean = 0.0f;
    float nearestHit = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
    {
        if (MIFX_R5_BATCH < 8 && s > 0 && s % MIFX_R5_BATCH == 0) // (the next batch, when a batch holds fewer than eight taps)
        {
#pragma unroll
            for (int t = 0; t < MIFX_R5_BATCH && s + t < 8; ++t) rayTexel[t] = ld<v4>(dirPdfTex, tapX[s + t], tapY[s + t]);
#pragma unroll
            for (int t = 0; t < MIFX_R5_BATCH && s + t < 8; ++t) colTexel[t] = ld<v4>(specTex, tapX[s + t], tapY[s + t]);
        }
        const float ws = spatial_weight_const(c_ssr_poisson[s][2] * c_ssr_poisson[s][2], 0.9f);
        // ComputeWeightRayLength :60-88
        const v4    dp  = rayTexel[s % MIFX_R5_BATCH];
        const float len = length(xyz(dp));
        const bool  ray = !(len < 1e-6f);
        float wgt, rayLen;
        {
            const v3    L = xyz(dp) / len;
            const v3    Hh = normalize(L + V);
            const float NdotH = saturate(dot(N, Hh)), NdotL = saturate(dot(N, L));
            // (the two GGX terms and the division by the pdf with the 1-ulp reciprocal / square root: smooth, cancellation-free -- mifx_pbr.h; L, H and the
            //  cosines above them stay on the strict path)
            const float vis = smith_ggx_visibility_correlated_v_q(NdotL, NdotV, alpha, visV);
            const float D   = normal_distribution_ggx_q(NdotH, alpha);
            float brdf = vis * D * NdotL;
            brdf *= ws;
            wgt    = ray ? fmaxf(brdf * q_rcp(fmaxf(dp.w, 1e-5f)), 1e-6f) : 1e-6f; // (no ray in the texel: the arithmetic above ran on a zero direction and is discarded)
            rayLen = ray ? len : 1e-6f;
        }
        const v4 c = colTexel[s % MIFX_R5_BATCH];
        // ComputeWeightedVariance :90-100
        colorSum  = colorSum + wgt * c;
        weightSum += wgt;
        const float value = luminance601(xyz(c));
        const float prevMean = mean;
        mean += wgt * q_rcp(weightSum) * (value - prevMean); // (a running mean: smooth in the weights)
        variance += wgt * (value - prevMean) * (value - mean);
        if (wgt > 1.0e-6f) nearestHit = fmaxf(rayLen, nearestHit);
    }
    const float invW = fdiv(1.0f, fmaxf(weightSum, 1e-6f)); // one division, five multiplies (the quotients move by <= 1.5 ulp)
    st<v4>(outRad, x, y, colorSum * invW);
    st<var_t>(outVar, x, y, variance * invW);
    // ComputeResolvedDepth :102-106
    st<var_t>(outDepth, x, y, camera_z_to_depth(length(camPos - posWS) + nearestHit, cam.proj));
}


// ------------------------------------------------------------------------------------------------ launchers
static const dim3 kBlock(64, 4, 1);
#define MIFX_LAUNCH_END()              \
    MIFX_HIP_CHECK(hipGetLastError()); \
    return MIFX_OK

mifx_status launch_ssr_hiz_pyramid(hipStream_t s, const Pyr& p, Img level0Copy, bool reversedDepth) // p.l[0] = depth; fills p.l[1 .. levels - 1] and the copy of level 0
{
    constexpr int kTailTexels = 16384; // levels of at most this many texels are left to the one-workgroup tail (whole-image levels only)
    bool copied = level0Copy.p == nullptr; // (no copy asked for: the march reads level 0 where it lies, HizSlab::base0)
    for (int k = 1; k < p.levels;)
    {
        const int nl = pyramid_fusable_levels(p.l[k - 1].w, p.l[k - 1].h, p.levels - k);
        if (k == 1 && nl < 2 && !copied)
        {
            MIFX_HIP_CHECK(hipMemcpy2DAsync(level0Copy.p, size_t(level0Copy.pitch), p.l[0].p, size_t(p.l[0].pitch), size_t(p.l[0].w) * 4u, size_t(p.l[0].h), hipMemcpyDeviceToDevice, s));
            copied = true;
        }
        if (nl >= 2)
        {
            HizOp op{};
            op.reversed = reversedDepth ? 1 : 0;
            op.src = p.l[k - 1];
            if (k == 1 && level0Copy.p != nullptr) { op.copy0 = level0Copy; copied = true; }
            op.pairs = pair_aligned(op.src) && (op.copy0.p == nullptr || pair_aligned(op.copy0)) ? 1 : 0;
            for (int j = 0; j < nl; ++j) op.dst[j] = p.l[k + j];
            hipLaunchKernelGGL(ssr_hiz_levels_kernel, dim3((p.l[k].w + 15) / 16, (p.l[k].h + 15) / 16, 1), dim3(256, 1, 1), 0, s, op, nl);
            k += nl;
        }
        else if (p.levels - k == 2 && p.l[k].yn == 0 && p.l[k + 1].yn == 0 && p.l[k].w * p.l[k].h <= (1 << 20))
        {
            const int midBlocks = (p.l[k].w * p.l[k].h + 255) / 256, lastBlocks = (p.l[k + 1].w * p.l[k + 1].h + 255) / 256;
            hipLaunchKernelGGL(ssr_hiz_last_two_kernel, dim3(unsigned(midBlocks + lastBlocks), 1, 1), dim3(256, 1, 1), 0, s, p.l[k - 1], p.l[k], p.l[k + 1], midBlocks, reversedDepth ? 1 : 0);
            k = p.levels;
        }
        else if (p.levels - k >= 2 && p.l[k].w * p.l[k].h <= kTailTexels && p.l[k].yn == 0)
        {
            HizTail t{};
            t.count = p.levels - k + 1;
            for (int j = 0; j < t.count; ++j) t.lv[j] = p.l[k - 1 + j];
            hipLaunchKernelGGL(ssr_hiz_tail_kernel, dim3(1, 1, 1), dim3(1024, 1, 1), 0, s, t, reversedDepth ? 1 : 0);
            k = p.levels;
        }
        else
        {
            hipLaunchKernelGGL(ssr_hiz_mip_kernel, grid2d(p.l[k], kBlock), kBlock, 0, s, p.l[k - 1], p.l[k], reversedDepth ? 1 : 0);
            ++k;
        }
        MIFX_HIP_CHECK(hipGetLastError());
    }
    if (!copied) MIFX_HIP_CHECK(hipMemcpy2DAsync(level0Copy.p, size_t(level0Copy.pitch), p.l[0].p, size_t(p.l[0].pitch), size_t(p.l[0].w) * 4u, size_t(p.l[0].h), hipMemcpyDeviceToDevice, s));
    return MIFX_OK;
}
mifx_status launch_ssr_mask_roughness(hipStream_t s, Img material, Img depth, Img roughness, Img mask, const mifx_ssr_attribs& a, bool reversedDepth)
{
    hipLaunchKernelGGL(ssr_mask_roughness_kernel, grid2d(mask, kBlock), kBlock, 0, s, material, depth, roughness, mask, make_k(a, reversedDepth));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_downsampled_mask(hipStream_t s, Img roughness, Img depth, Img mask, const mifx_ssr_attribs& a, bool reversedDepth)
{
    hipLaunchKernelGGL(ssr_downsampled_mask_kernel, grid2d(mask, kBlock), kBlock, 0, s, roughness, depth, mask, make_k(a, reversedDepth, true));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_spatial(hipStream_t s, Img roughness, Img normal, Img depth, Img dirPdf, Img spec, Img mask, Img outRad, Img outVar, Img outDepth, const CamK& cam,
                               const mifx_ssr_attribs& a, bool halfResolution)
{
    const SsrK k = make_k(a, cam.reversedDepth != 0, halfResolution);
    if (halfResolution) hipLaunchKernelGGL(ssr_spatial_kernel<true>, grid2d(outRad, kBlock), kBlock, 0, s, roughness, normal, depth, dirPdf, spec, mask, outRad, outVar, outDepth, cam, k);
    else hipLaunchKernelGGL(ssr_spatial_kernel<false>, grid2d(outRad, kBlock), kBlock, 0, s, roughness, normal, depth, dirPdf, spec, mask, outRad, outVar, outDepth, cam, k);
    MIFX_LAUNCH_END();
}
} // namespace mifx
