// ssao.hip -- ScreenSpaceAmbientOcclusion passes A2..A8 (XeGTAO-style GTAO / HBAO / visibility-bitmask AO with temporal
// accumulation, history-fix resampling and spatial denoise).  Math follows
// Shaders/PostProcess/ScreenSpaceAmbientOcclusion/private/SSAO_*.fx; the host sequence is in api_ssao.cpp.
//
// All planes are fp32 (AO, history length, depth).  Bandwidth accounting per pass: SURVEY.md Appendix C.
#include "mifx_host.h"
#include "mifx_effects.h"
#include "mifx_pyramid.h"

namespace mifx
{

#define SSAO_SLICE_COUNT 3
#define SSAO_SAMPLES_PER_SLICE 3
#define SSAO_MAX_MIP 4
#define M_PI_F 3.14159265358979f
#define M_HALF_PI_F 1.57079632679490f

// SSAO_Common.fxh:25-28
MIFX_D float geometry_weight(v3 centerPos, v3 tapPos, v3 centerNormal, float planeDistNorm)
{
    return saturate(1.0f - fabsf(dot(tapPos - centerPos, centerNormal)) * planeDistNorm);
}

// ------------------------------------------------------------------------------------------------ A2: prefiltered depth mip (SSAO_ComputePrefilteredDepthBuffer.fx:42-121)
__global__ __launch_bounds__(256) void ssao_prefilter_mip_kernel(Img src, Img dst, m44 proj, SsaoK k, int q16)
{
    int x, y;
    if (!pixel_xy(dst, x, y)) return;
    const int  rx = 2 * x, ry = 2 * y;
    const bool oddW = (src.w & 1) != 0, oddH = (src.h & 1) != 0;
    float s[9];
    int   n = 0;
    auto  tap = [&](int ox, int oy) { s[n++] = depth_to_camera_z(ld_clamp<float>(src, rx + ox, ry + oy), proj); };
    tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);

    // ComputeDepthMIPFiltered (:42-71): weighted average that favours the closest sample
    float wd = s[0];
    for (int i = 1; i < n; ++i) wd = fminf(wd, s[i]);
    const float effectRadius = 0.75f * k.EffectRadius * k.RadiusMultiplier;
    const float falloffRange = k.EffectFalloffRange * effectRadius;
    const float falloffFrom  = effectRadius - falloffRange;
    const float falloffMul   = fdiv(-1.0f, falloffRange);
    const float falloffAdd   = fdiv(falloffFrom, falloffRange) + 1.0f;
    float depthSum = 0.0f, weightSum = 0.0f;
    for (int i = 0; i < n; ++i)
    {
        float w = saturate(fabsf(wd - s[i]) * falloffMul + falloffAdd);
        depthSum += w * s[i];
        weightSum += w;
    }
    st<float>(dst, x, y, depth16(saturate(camera_z_to_depth(fdiv(depthSum, weightSum), proj)), q16)); // q16: the level is an R16_UNORM target (mifx_device.h: depth16)
}

struct PrefilterOp // A2 on an even-sized source: the four-tap case of ssao_prefilter_mip_kernel
{
    using T = float;
    Img   src, dst[4], zsrc, zdst[4]; // zsrc / zdst: camera-z of the source level and of every produced level
    m44   proj;
    float falloffMul, falloffAdd;
    int   q16; // the levels are R16_UNORM targets (FEATURE_FLAG_HALF_PRECISION_DEPTH in the native-storage build; the source level then already holds such values)
    // (zsrc may carry a row window, Img::y0 / yn, on an even boundary: row-band sharding writes the camera z of the source level only where its taps can reach)
    MIFX_D bool zrow(int y) const { return y >= zsrc.y0 && y < row_end(zsrc); }
    MIFX_D float load(int x, int y) const
    {
        const float d = ld<float>(src, x, y);
        if (zrow(y)) st<float>(zsrc, x, y, depth_to_camera_z(d, proj)); // every source texel is read by exactly one thread (even dimensions)
        return d;
    }
    int pairs; // src and zsrc allow 8-byte accesses (pair_aligned)
    MIFX_D void quad(int x, int y, float& a, float& b, float& c, float& d) const
    {
        if (pairs)
        {
            const v2 r0 = ld_pair(src, 2 * x, 2 * y), r1 = ld_pair(src, 2 * x, 2 * y + 1);
            if (zrow(2 * y))
            {
                st_pair(zsrc, 2 * x, 2 * y, v2{depth_to_camera_z(r0.x, proj), depth_to_camera_z(r0.y, proj)});
                st_pair(zsrc, 2 * x, 2 * y + 1, v2{depth_to_camera_z(r1.x, proj), depth_to_camera_z(r1.y, proj)});
            }
            a = r0.x; b = r1.x; c = r0.y; d = r1.y;
        }
        else { a = load(2 * x, 2 * y); b = load(2 * x, 2 * y + 1); c = load(2 * x + 1, 2 * y); d = load(2 * x + 1, 2 * y + 1); }
    }
    MIFX_D float reduce(float d0, float d1, float d2, float d3) const
    {
        const float s[4] = {depth_to_camera_z(d0, proj), depth_to_camera_z(d1, proj), depth_to_camera_z(d2, proj), depth_to_camera_z(d3, proj)};
        const float wd   = fminf(fminf(fminf(s[0], s[1]), s[2]), s[3]);
        float depthSum = 0.0f, weightSum = 0.0f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
        {
            const float w = saturate(fabsf(wd - s[i]) * falloffMul + falloffAdd);
            depthSum += w * s[i];
            weightSum += w;
        }
        return saturate(camera_z_to_depth(fdiv(depthSum, weightSum), proj));
    }
    MIFX_D float stored(float v) const { return depth16(v, q16); }
    // Row windows on dst / zdst (Img::y0 / yn; row-band sharding) bound what is STORED: every level is still reduced whole -- the last one is read anywhere -- but a
    // rank's A3 reads level k only within 2^(k + 0.5 + DepthMIPSamplingOffset) pixels of its own rows (tap_mip_offset), and nothing else reads these levels.
    MIFX_D bool inside(int l, int x, int y) const { return x < dst[l - 1].w && y < dst[l - 1].h; }
    // Round 6: a row window on the SOURCE (src.y0 / yn, a multiple of 2^nl rows) bounds what is REDUCED -- a rank of a sharded frame that gets the rows of the last level it
    // does not own from their owners (api_comm.cpp: the level-4 all-gather) reduces only the source rows its own store windows and its own rows of the last level need.
    MIFX_D int  first_row() const { return src.y0 >> 1; }
    MIFX_D void store(int l, int x, int y, float v) const
    {
        if (y < dst[l - 1].y0 || y >= row_end(dst[l - 1])) return;
        v = depth16(v, q16);
        st<float>(dst[l - 1], x, y, v);
        st<float>(zdst[l - 1], x, y, depth_to_camera_z(v, proj)); // what a consumer would compute from the stored depth
    }
};
// camera-z of one pyramid level (generic path: odd-sized sources)
__global__ __launch_bounds__(256) void ssao_depth_to_camz_kernel(Img depth, Img camz, m44 proj)
{
    int x, y;
    if (!pixel_xy(camz, x, y)) return;
    st<float>(camz, x, y, depth_to_camera_z(ld<float>(depth, x, y), proj));
}
__global__ __launch_bounds__(256) void ssao_prefilter_levels_kernel(PrefilterOp op, int nl) { pyramid_reduce_levels(op, nl); }

// ------------------------------------------------------------------------------------------------ A1: checkerboard depth (SSAO_ComputeDownsampledDepth.fx:8-28; half resolution)
__global__ __launch_bounds__(256) void ssao_downsample_depth_kernel(Img depth, Img out)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const float d0 = ld_zero_f(depth, 2 * x, 2 * y), d1 = ld_zero_f(depth, 2 * x, 2 * y + 1), d2 = ld_zero_f(depth, 2 * x + 1, 2 * y), d3 = ld_zero_f(depth, 2 * x + 1, 2 * y + 1);
    const float mn = fminf(fminf(d0, d1), fminf(d2, d3)), mx = fmaxf(fmaxf(d0, d1), fmaxf(d2, d3));
    st<float>(out, x, y, lerpf(mn, mx, float((x + y) & 1))); // ComputeCheckerboardPattern
}

// ------------------------------------------------------------------------------------------------ A4: bilateral upsampling (SSAO_ComputeBilateralUpsampling.fx:66-139; half resolution)
__global__ __launch_bounds__(256) void ssao_bilateral_upsample_kernel(Img depth, Img occl, Img out, CamK cam)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const float center = ld<float>(depth, x, y);
    if (is_background(center, cam.reversedDepth != 0))
    {
        st<ao_t>(out, x, y, 1.0f);
        return;
    }
    const int   hw = int(0.5f * cam.vw), hh = int(0.5f * cam.vh); // int2(0.5 * f4ViewportSize.xy)
    const int   cx = int(0.5f * float(x)), cy = int(0.5f * float(y)); // int2(0.5 * floor(Position))
    const float z0 = depth_to_camera_z(center, cam.proj), invZ0 = fmaxf(z0, 1e-6f);
    float sum = 0.0f, wsum = 0.0f;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
        {
            const int   lx = clampi(cx + dx, 0, hw - 1), ly = clampi(cy + dy, 0, hh - 1); // ClampScreenCoord
            const float u = 2.0f * (float(lx) + 0.5f) * cam.ivw, v = 2.0f * (float(ly) + 0.5f) * cam.ivh;
            const float signal = ld_zero_f<ao_t>(occl, lx, ly);
            const float guide  = sample_linear_clamp_f(depth, u, v);
            const float ws = spatial_weight_const(float(dx * dx + dy * dy), 0.9f); // SSAO_BILATERAL_UPSAMPLING_SIGMA
            const float alpha = fdiv(fabsf(z0 - depth_to_camera_z(guide, cam.proj)), invZ0); // ComputeDepthWeight :66-72
            // SSAO_BILATERAL_UPSAMPLING_DEPTH_SIGMA.  Not the hardware exp: the weights of a pixel across a depth edge are denormal (exp(-90)), and
            // "WeightSum > 0" below decides between them and the fallback.  The last step of that range decides too: exp(-103.8) is 0.6 of the
            // smallest denormal, which a correctly rounded expf returns as that denormal (weight sum > 0, the ratio of two one-unit numbers = 1.0)
            // and the device library's expf returns as zero (fallback) -- whole rows of a floor seen at a glancing angle sit there.  The double
            // exponential, rounded once to float, is the correctly rounded value; this pass only runs in the half-resolution mode
            const float wz = float(exp(double(fdiv(-(alpha * alpha), 2.0f * 0.0075f * 0.0075f))));
            sum += ws * wz * signal;
            wsum += ws * wz;
        }
    // (IEEE division: the weight sum may be a denormal number, which fdiv's reciprocal does not handle)
    st<ao_t>(out, x, y, wsum > 0.0f ? sum / wsum : sample_linear_clamp_f<ao_t>(occl, 2.0f * (float(cx) + 0.5f) * cam.ivw, 2.0f * (float(cy) + 0.5f) * cam.ivh));
}

struct ConvoluteOp // A6 on even-sized sources: x = AO, y = depth
{
    using T = v2;
    Img srcAO, srcDepth, dstAO[4], dstDepth[4];
    int pairs; // srcAO and srcDepth allow 8-byte accesses (pair_aligned)
    int q16;   // the depth levels are R16_UNORM targets (see PrefilterOp)
    MIFX_D v2   load(int x, int y) const { return v2{ld<ao_t>(srcAO, x, y), ld<float>(srcDepth, x, y)}; }
    MIFX_D void quad(int x, int y, v2& a, v2& b, v2& c, v2& d) const
    {
        if (pairs)
        {
            const v2 a0 = ld_pair(srcAO, 2 * x, 2 * y), a1 = ld_pair(srcAO, 2 * x, 2 * y + 1), d0 = ld_pair(srcDepth, 2 * x, 2 * y), d1 = ld_pair(srcDepth, 2 * x, 2 * y + 1);
            a = v2{a0.x, d0.x}; b = v2{a1.x, d1.x}; c = v2{a0.y, d0.y}; d = v2{a1.y, d1.y};
        }
        else { a = load(2 * x, 2 * y); b = load(2 * x, 2 * y + 1); c = load(2 * x + 1, 2 * y); d = load(2 * x + 1, 2 * y + 1); }
    }
    MIFX_D v2   reduce(v2 a, v2 b, v2 c, v2 d) const { return v2{(((a.x + b.x) + c.x) + d.x) * 0.25f, (((a.y + b.y) + c.y) + d.y) * 0.25f}; } // sum / 4
    MIFX_D v2   stored(v2 v) const { return v2{quantize_as<ao_t>(v.x), depth16(v.y, q16)}; }
    MIFX_D bool inside(int l, int x, int y) const { return x < dstAO[l - 1].w && y < row_end(dstAO[l - 1]); }
    MIFX_D int  first_row() const { return dstAO[0].y0; } // (a multiple of 8: mifx_ssao::kWindowAlign rows of the frame)
    MIFX_D void store(int l, int x, int y, v2 v) const { st<ao_t>(dstAO[l - 1], x, y, v.x); st<float>(dstDepth[l - 1], x, y, depth16(v.y, q16)); }
};
__global__ __launch_bounds__(256) void ssao_convolute_levels_kernel(ConvoluteOp op, int nl) { pyramid_reduce_levels(op, nl); }

// ------------------------------------------------------------------------------------------------ A5: temporal accumulation (SSAO_ComputeTemporalAccumulation.fx:76-180)
// one texel of the pass: stores the accumulated AO and the history length, returns them (x = AO, y = history length)
MIFX_D v2 ssao_temporal_texel(int x, int y, const Img& currAO, const Img& prevAO, const Img& prevLen, const Img& currDepth /*reprojected*/, const Img& prevDepth, const Img& motionTex,
                              const Img& outAO, const Img& outLen, const CamK& cur, const CamK& prev, const SsaoK& k)
{
    // Memory-level parallelism (round 3): depth and motion are fetched together, and everything that depends on the reprojected position -- the four history depths,
    // the four history AO / length texels and the 3x3 neighbourhood of the current AO -- is in flight before the depth-similarity test that decides whether it is
    // used (two round trips instead of five; a pixel that fails the test has fetched 21 texels from cache for nothing).  Same arithmetic on the same values.
    float depth = ld<float>(currDepth, x, y);
    v2    m     = ld<cm_t>(motionTex, x, y);
    keep_here(depth); keep_here(m); // (both arrive together: the motion vector is not fetched behind the background test)
    if (is_background(depth, cur.reversedDepth != 0))
    {
        st<ao_t>(outAO, x, y, 1.0f); // discard: both targets keep their cleared value 1.0 (.cpp:1059-1068)
        st<hl_t>(outLen, x, y, 1.0f);
        return v2{1.0f, 1.0f};
    }
    const v2 motion{m.x * 0.5f, m.y * -0.5f};
    const v2 prevLoc{(float(x) + 0.5f) - motion.x * cur.vw, (float(y) + 0.5f) - motion.y * cur.vh};

    // ComputeReprojection :105-149
    const float    currCamZ = depth_to_camera_z(depth, cur.proj);
    const int      W = int(cur.vw), H = int(cur.vh);
    const Bilinear b = bilinear_uc(prevLoc.x, prevLoc.y, W, H);
    const v4 pd{ld<float>(prevDepth, b.x0, b.y0), ld<float>(prevDepth, b.x1, b.y0), ld<float>(prevDepth, b.x0, b.y1), ld<float>(prevDepth, b.x1, b.y1)};
    const v4 po{ld<ao_t>(prevAO, b.x0, b.y0), ld<ao_t>(prevAO, b.x1, b.y0), ld<ao_t>(prevAO, b.x0, b.y1), ld<ao_t>(prevAO, b.x1, b.y1)};
    v4       h{ld<hl_t>(prevLen, b.x0, b.y0), ld<hl_t>(prevLen, b.x1, b.y0), ld<hl_t>(prevLen, b.x0, b.y1), ld<hl_t>(prevLen, b.x1, b.y1)};
    float nb[9]; // the 3x3 neighbourhood of the current AO, (dx, dy) in the order of the statistic's loops; nb[4] is the pixel itself
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) nb[(dx + 1) * 3 + (dy + 1)] = ld<ao_t>(currAO, clampi(x + dx, 0, W - 1), clampi(y + dy, 0, H - 1));
    auto similar = [&](float d) __attribute__((always_inline)) {
        float pz = depth_to_camera_z(d, prev.proj);
        return fabsf(1.0f - fdiv(currCamZ, pz)) < 0.01f ? 1.0f : 0.0f; // IsCameraZSimilar :76-79, SSAO_DISOCCLUSION_DEPTH_THRESHOLD
    };
    v4 w{b.w00 * similar(pd.x), b.w10 * similar(pd.y), b.w01 * similar(pd.z), b.w11 * similar(pd.w)};
    const float totalW = dot(w, mk4(1.0f));
    const bool success = totalW > 0.01f && !k.ResetAccumulation;
    // (evaluated for every pixel and selected: a branch here would let the compiler sink the fetches above into it, behind the test -- a pixel that fails the test
    //  divides by a weight sum near zero and the select discards the result)
    float occ, hist;
    {
        h    = min4(h + mk4(1.0f), mk4(16.0f)); // SSAO_MAX_HISTORY_LENGTH
        occ  = fdiv(dot(po, w), totalW);
        hist = fdiv(dot(h, w), totalW);

        // ComputePixelStatistic :81-103 (3x3, clamped)
        float m1 = 0.0f, m2 = 0.0f;
#pragma unroll
        for (int i = 0; i < 9; ++i)
        {
            const float s = nb[i];
            m1 += s;
            m2 += s * s;
        }
        const float mean = fdiv(m1, 9.0f);
        const float var  = fdiv(m2, 9.0f) - (mean * mean);
        const float sd   = fsqrt(fmaxf(var, 0.0f));
        const float aspect = cur.vw * cur.ivh;
        const float motionFactor  = saturate(1.025f - length(v2{motion.x * aspect, motion.y}) * 128.0f); // SSAO_TEMPORAL_MOTION_VECTOR_DIFF_FACTOR
        const float varianceGamma = lerpf(0.5f, 2.5f, motionFactor * motionFactor);
        const float omin = mean - varianceGamma * sd, omax = mean + varianceGamma * sd;
        const bool  inside = omin < occ && occ < omax;
        hist = inside ? hist : fmaxf(1.0f, motionFactor * hist);
    }
    occ  = success ? occ : 1.0f;
    hist = success ? hist : 1.0f;
    const float alpha = fdiv(1.0f, hist);
    const float ao    = lerpf(occ, nb[4], alpha);
    st<ao_t>(outAO, x, y, ao);
    st<hl_t>(outLen, x, y, hist);
    return v2{ao, hist};
}

// History length -> the two "enough history" measures of the resolve: A7 copies the accumulated AO when (hist - 1) / 4 >= 1 (SSAO_ComputeResampledHistory.fx:61-64,
// SSAO_OCCLUSION_HISTORY_MAX_FRAMES_WITH_HISTORY_FIX), A8 skips its filter when pow(|hist - 1| / 8, 0.2) >= 1 (SSAO_ComputeSpatialReconstruction.fx:52-60,
// SSAO_OCCLUSION_HISTORY_MAX_FRAMES_WITH_DENOISING).
MIFX_D float ssao_resample_accum(float hist) { return (hist - 1.0f) / 4.0f; }
MIFX_D float ssao_spatial_accum(float hist) { return m_pow(fabsf((hist - 1.0f) / 8.0f), 0.2f); }

// ---- A7 + A8 as one resolve folded into A5 ("fused resolve", the default; the full-frame passes further down remain behind mifx_debug_ssao_set_fused_resolve).
// At a texel with enough history A7 is a copy (`return LoadOcclusion`, SSAO_ComputeResampledHistory.fx:61-64) and A8 a lerp (SSAO_ComputeSpatialReconstruction.fx:56-60)
// of values A5 holds in registers at that moment.  The temporal pass therefore also
//   * writes A7's copy (resampled := accumulated AO) for every texel,
//   * writes the final value lerp(1, accumulated AO, AlphaInterpolation) where A8 takes its early path -- background, or pow(|hist - 1| / 8, 0.2) >= 1; for such a texel
//     A7's early-out holds as well ((hist - 1) / 4 >= 1 follows from |hist - 1| / 8 >= 1/2, far inside the error of the hardware pow),
//   * appends every other texel to the `spatial` list, and to the `walk` list too when A7 resamples it (not background, (hist - 1) / 4 < 1).
// Two work-list passes follow: A7's pyramid walk (ssao_resample_walk) overwrites the copy at the texels of the walk list; A8's filter (ssao_spatial_filter) then runs
// at the texels of the spatial list on the completed resampled plane.  Every texel gets the bits the two full-frame passes give it
// (tests/test_gpu_ssao.py: test_ssao_fused_resolve_is_bit_identical); the two full-frame passes over hist / depth / AO (and A7's store of the untouched texels) go away.
// Lists without atomics: the workgroup b of this launch owns the entries [256 b, 256 b + count_b) of each list (its 64x4 texels in row-major order) and writes
// count_b; the list passes walk the segments.
struct ResolveOut
{
    Img       depthTex;            // the depth buffer of A7 / A8's background test (A5 itself reads the reprojected depth)
    Img       resampled;           // row window: the rows A8's taps can reach
    Img       out, out2;           // row window: the rows of the output; out2 optional (the stable output plane beside the history plane)
    float     alphaInterpolation;
    unsigned* counts;              // [2 b] walk, [2 b + 1] spatial
    unsigned* walk;                // y << 16 | x
    unsigned* spatial;
};
MIFX_D unsigned lanes_below(unsigned long long m) { return __builtin_amdgcn_mbcnt_hi(unsigned(m >> 32), __builtin_amdgcn_mbcnt_lo(unsigned(m), 0u)); } // set bits of m below this lane
MIFX_D bool in_rows(const Img& im, int y) { return y >= im.y0 && y < row_end(im); }

template <bool RESOLVE> __global__ __launch_bounds__(256) void ssao_temporal_kernel(Img currAO, Img prevAO, Img prevLen, Img currDepth /*reprojected*/, Img prevDepth, Img motionTex, Img outAO,
                                                                             Img outLen, CamK cur, CamK prev, SsaoK k, ResolveOut R)
{
    int x, y;
    const bool in = pixel_xy_dir<32>(outAO, x, y);
    if (!RESOLVE)
    {
        if (in) ssao_temporal_texel(x, y, currAO, prevAO, prevLen, currDepth, prevDepth, motionTex, outAO, outLen, cur, prev, k);
        return;
    }
    __shared__ unsigned waveCount[2][4];
    bool walk = false, spatial = false;
    if (in)
    {
        const float d8 = ld<float>(R.depthTex, x, y); // (first: in flight with the pass's own depth and motion)
        const v2    r  = ssao_temporal_texel(x, y, currAO, prevAO, prevLen, currDepth, prevDepth, motionTex, outAO, outLen, cur, prev, k);
        const float ao = quantize_as<ao_t>(r.x), hist = quantize_as<hl_t>(r.y); // what A7 / A8 would load back from the planes
        const bool  bg   = is_background(d8, cur.reversedDepth != 0);
        const bool  done = bg || ssao_spatial_accum(hist) >= 1.0f;
        const bool  a7   = in_rows(R.resampled, y), a8 = in_rows(R.out, y);
        walk    = a7 && !bg && ssao_resample_accum(hist) < 1.0f;
        spatial = a8 && !done;
        if (a7) st<ao_t>(R.resampled, x, y, ao);
        if (a8 && done)
        {
            const float v = lerpf(1.0f, ao, R.alphaInterpolation);
            st<ao_t>(R.out, x, y, v);
            if (R.out2.p) st<ao_t>(R.out2, x, y, v);
        }
    }
    // segment append (block 64 x 4 = four waves, one row each)
    const unsigned wave = threadIdx.y, block = blockIdx.y * gridDim.x + blockIdx.x;
    const unsigned long long mW = __ballot(walk), mS = __ballot(spatial);
    if (threadIdx.x == 0u) { waveCount[0][wave] = unsigned(__popcll(mW)); waveCount[1][wave] = unsigned(__popcll(mS)); }
    __syncthreads();
    unsigned baseW = 0u, baseS = 0u, totalW = 0u, totalS = 0u;
#pragma unroll
    for (unsigned w = 0; w < 4u; ++w)
    {
        const unsigned cw = waveCount[0][w], cs = waveCount[1][w];
        if (w < wave) { baseW += cw; baseS += cs; }
        totalW += cw; totalS += cs;
    }
    const unsigned item = (unsigned(y) << 16) | unsigned(x);
    if (walk) R.walk[block * 256u + baseW + lanes_below(mW)] = item;
    if (spatial) R.spatial[block * 256u + baseS + lanes_below(mS)] = item;
    if (threadIdx.x == 0u && threadIdx.y == 0u) { R.counts[2u * block] = totalW; R.counts[2u * block + 1u] = totalS; }
}

// ------------------------------------------------------------------------------------------------ A6: convoluted AO-history / depth pyramids (SSAO_ComputeConvolutedDepthHistory.fx:41-110)
__global__ __launch_bounds__(256) void ssao_convolute_mip_kernel(Img srcAO, Img srcDepth, Img dstAO, Img dstDepth, int q16)
{
    int x, y;
    if (!pixel_xy(dstAO, x, y)) return;
    const int  rx = 2 * x, ry = 2 * y;
    const bool oddW = (srcAO.w & 1) != 0, oddH = (srcAO.h & 1) != 0;
    float a = 0.0f, d = 0.0f;
    int   n = 0;
    auto  tap = [&](int ox, int oy) { a += ld_clamp<ao_t>(srcAO, rx + ox, ry + oy); d += ld_clamp<float>(srcDepth, rx + ox, ry + oy); ++n; };
    tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
    if (oddW) { tap(2, 0); tap(2, 1); }
    if (oddH) { tap(0, 2); tap(1, 2); }
    if (oddW && oddH) tap(2, 2);
    st<ao_t>(dstAO, x, y, fdiv(a, float(n)));
    st<float>(dstDepth, x, y, depth16(fdiv(d, float(n)), q16));
}

// ------------------------------------------------------------------------------------------------ A7: resampled history (SSAO_ComputeResampledHistory.fx:56-113)
// EXACT: the frame is divisible by 16 in both directions, so MipResolution = viewport / 2^mip IS the size of level mip and the four taps -- taken at
// (texel + 0.5) / MipResolution -- sit on texel centres: the linear-clamp sample of the depth is that texel (the checker's fp32 bilinear weights leave it 1 - O(1e-5) and
// its neighbour the rest: 1e-5 of a depth difference between adjacent texels of a box-filtered level -- a stated deviation of ~1e-5 relative, not bit-identity), the
// point sample of the AO is the same texel.  Two clamped
// loads instead of a bilinear tap (62 instructions) and a point tap per sample -- a third of the slow path; other frame sizes keep the general taps.
//
// The pyramid walk of one pixel whose history is shorter than SSAO_OCCLUSION_HISTORY_MAX_FRAMES_WITH_HISTORY_FIX (:66-113); shared by the full-frame pass
// (ssao_resample_kernel) and the work-list pass of the fused resolve (ssao_resample_list_kernel): the same instructions, so the two produce the same bits.
template <bool EXACT> MIFX_D float ssao_resample_walk(int x, int y, float depth, float accum, const Img* aoLv, const Img* depthLv, const Img& normal, const CamK& cam)
{
    int      mip = int(4.0f * (1.0f - saturate(accum))); // SSAO_DEPTH_HISTORY_CONVOLUTED_MAX_MIP
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    const v3 positionVS = screen_xy_depth_to_view_space(v3{pos.x * cam.ivw, pos.y * cam.ivh, depth}, cam.proj);
    const v3 normalVS   = mul_dir(xyz(ld<v4>(normal, x, y)), cam.view);
    const float planeNormalFactor = fdiv(10.0f, 1.0f + depth_to_camera_z(depth, cam.proj));

    float occSum = 0.0f, wSum = 0.0f;
    while (mip >= 0 && wSum < 0.995f)
    {
        const float inv = fdiv(1.0f, float(1u << unsigned(mip)));
        const v2    mipRes{cam.vw * inv, cam.vh * inv};
        const v2    mipLoc{pos.x * inv, pos.y * inv};
        const v2    invMipRes{fdiv(1.0f, mipRes.x), fdiv(1.0f, mipRes.y)}; // rcp(MipResolution), hoisted out of the 4-tap loop
        const int   lx = int(mipLoc.x - 0.5f), ly = int(mipLoc.y - 0.5f); // int(): truncation toward zero, as in HLSL
        const float fx = fracf(mipLoc.x + 0.5f), fy = fracf(mipLoc.y + 0.5f);
        const float wgt[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
        occSum = 0.0f;
        wSum   = 0.0f;
        // (the eight texels of the level's four taps are requested together, then consumed in the reference's order: one round trip per level instead of four)
        float sd[4], so[4];
        v2    tcs[4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            const int sx = lx + (s & 1), sy = ly + (s >> 1);
            tcs[s] = v2{(float(sx) + 0.5f) * invMipRes.x, (float(sy) + 0.5f) * invMipRes.y};
            sd[s]  = EXACT ? ld_clamp<float>(depthLv[mip], sx, sy) : sample_linear_clamp_f(depthLv[mip], tcs[s].x, tcs[s].y); // Sam_LinearClamp (.cpp:735)
            so[s]  = EXACT ? ld_clamp<ao_t>(aoLv[mip], sx, sy) : sample_point_clamp_f<ao_t>(aoLv[mip], tcs[s].x, tcs[s].y);       // Sam_PointClamp  (.cpp:736)
        }
#pragma unroll
        for (int s = 0; s < 4; ++s)
        {
            const v3    sampleVS = screen_xy_depth_to_view_space(v3{tcs[s].x, tcs[s].y, sd[s]}, cam.proj);
            const float ws = wgt[s];
            const float wz = geometry_weight(positionVS, sampleVS, normalVS, planeNormalFactor);
            occSum += so[s] * ws * wz;
            wSum += ws * wz;
        }
        --mip;
    }
    return fdiv(occSum, wSum);
}
template <bool EXACT> __global__ __launch_bounds__(256) void ssao_resample_kernel(Pyr aoPyr, Pyr depthPyr, Img histLen, Img normal, Img out, CamK cam)
{
    __shared__ Img aoLv[8], depthLv[8];
    {
        const unsigned t = threadIdx.y * blockDim.x + threadIdx.x;
        if (t < 8u) aoLv[t] = aoPyr.l[t];
        else if (t < 16u) depthLv[t - 8u] = depthPyr.l[t - 8u];
        __syncthreads();
    }
    int x, y;
    const bool inWindow = tiled_xy(out, x, y); // divergent per-pixel loop (only recently disoccluded pixels resample): 8x8 wave tiles cut the number of waves a silhouette touches
    if (!inWindow) return;
    const float depth = ld<float>(depthPyr.l[0], x, y);
    const float hist  = ld<hl_t>(histLen, x, y);
    const float accum = ssao_resample_accum(hist);
    if (is_background(depth, cam.reversedDepth != 0) || accum >= 1.0f)
    {
        st<ao_t>(out, x, y, ld<ao_t>(aoPyr.l[0], x, y));
        return;
    }
    st<ao_t>(out, x, y, ssao_resample_walk<EXACT>(x, y, depth, accum, aoLv, depthLv, normal, cam));
}

// ------------------------------------------------------------------------------------------------ A8: spatial reconstruction (SSAO_ComputeSpatialReconstruction.fx:43-108) + history write-back
static constexpr float c_poisson[8][3] = {{-0.4706069f, -0.4427112f, +0.6461146f}, {-0.9057375f, +0.3003471f, +0.9542373f}, {-0.3487388f, +0.4037880f, +0.5335386f},
                                      {+0.1023042f, +0.6439373f, +0.6520134f}, {+0.5699277f, +0.3513750f, +0.6695386f}, {+0.2939128f, -0.1131226f, +0.3149309f},
                                      {+0.7836658f, -0.4208784f, +0.8895339f}, {+0.1564120f, -0.8198990f, +0.8346850f}};

// The eight-tap filter of a pixel without enough history (:62-100).  `resampled(sx, sy)` returns A7's value of a texel: a load of the resampled plane in the
// full-frame pass, the rule of the fused resolve (below) in the work-list pass.
template <class RESAMPLED> MIFX_D float ssao_spatial_filter(int x, int y, float accum, const Img& camzTex, const Img& normal, const CamK& cam, const SsaoK& k, RESAMPLED resampled)
{
    const v2 pos{float(x) + 0.5f, float(y) + 0.5f};
    const float camz    = ld<float>(camzTex, x, y); // == depth_to_camera_z(depth, proj), written by A2
    const v3 positionVS = screen_xy_camz_to_view_space(pos.x * cam.ivw, pos.y * cam.ivh, camz, cam.proj);
    const v3 normalVS   = mul_dir(xyz(ld<v4>(normal, x, y)), cam.view);
    const float angle   = 2.0f * M_PI_F * bayer4x4(unsigned(x), unsigned(y), cam.frameIndex);
    float sinA, cosA;
    m_sincos(angle, sinA, cosA); // angle in [0, 2 pi)
    const v4 rot{cosA, sinA, -sinA, cosA}; // GetRotator (PostFX_Common.fxh:67-73)
    const float radius = lerpf(0.0f, k.SpatialReconstructionRadius, 1.0f - saturate(accum));
    const float planeNormalFactor = fdiv(10.0f, 1.0f + camz);
    const int   W = int(cam.vw), H = int(cam.vh);
    float occSum = 0.0f, wSum = 0.0f;
#pragma unroll
    for (int s = 0; s < 8; ++s)
    {
        const v2  xi = rotate_vector(rot, v2{c_poisson[s][0], c_poisson[s][1]});
        const int sx = clampi(int(pos.x + radius * xi.x), 0, W - 1), sy = clampi(int(pos.y + radius * xi.y), 0, H - 1);
        const float sz = ld<float>(camzTex, sx, sy);
        const float so = resampled(sx, sy);
        const v3 sampleVS = screen_xy_camz_to_view_space((float(sx) + 0.5f) * cam.ivw, (float(sy) + 0.5f) * cam.ivh, sz, cam.proj);
        const float ws = spatial_weight_const(c_poisson[s][2] * c_poisson[s][2], 0.9f); // SSAO_SPATIAL_RECONSTRUCTION_SIGMA
        const float wz = geometry_weight(positionVS, sampleVS, normalVS, planeNormalFactor);
        occSum += ws * wz * so;
        wSum += ws * wz;
    }
    const float o = wSum > 0.0f ? fdiv(occSum, wSum) : resampled(x, y);
    return lerpf(1.0f, o, k.AlphaInterpolation);
}

__global__ __launch_bounds__(256) void ssao_spatial_kernel(Img occl, Img histLen, Img depthTex, Img camzTex, Img normal, Img out, Img historyOut, CamK cam, SsaoK k)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const float hist  = ld<hl_t>(histLen, x, y);
    const float depth = ld<float>(depthTex, x, y);
    const float accum = ssao_spatial_accum(hist);
    float result;
    if (is_background(depth, cam.reversedDepth != 0) || accum >= 1.0f) result = lerpf(1.0f, ld<ao_t>(occl, x, y), k.AlphaInterpolation);
    else result = ssao_spatial_filter(x, y, accum, camzTex, normal, cam, k, [&](int sx, int sy) __attribute__((always_inline)) { return ld<ao_t>(occl, sx, sy); });
    st<ao_t>(out, x, y, result);
    if (historyOut.p) st<ao_t>(historyOut, x, y, result); // CopyTexture resolved -> history[curr] (ScreenSpaceAmbientOcclusion.cpp:1319-1328), fused
}

// ------------------------------------------------------------------------------------------------ the work-list passes of the fused resolve (lists: ssao_temporal_kernel<true>)
// A workgroup takes four consecutive segments at a time and spreads their entries over its lanes (a silhouette leaves a few entries in many segments).
struct SegmentCursor
{
    unsigned first, n[4], total;
    MIFX_D SegmentCursor(const unsigned* counts, unsigned which, unsigned seg0, unsigned nseg) : first(seg0), total(0u)
    {
#pragma unroll
        for (unsigned j = 0; j < 4u; ++j)
        {
            n[j] = seg0 + j < nseg ? counts[2u * (seg0 + j) + which] : 0u;
            total += n[j];
        }
    }
    MIFX_D unsigned entry(const unsigned* list, unsigned i) const // the i-th entry of the four segments, i < total
    {
        unsigned j = 0u;
#pragma unroll
        for (unsigned t = 0; t < 3u; ++t)
            if (j == t && i >= n[t]) { i -= n[t]; j = t + 1u; }
        return list[(first + j) * 256u + i];
    }
};
template <bool EXACT> __global__ __launch_bounds__(256) void ssao_resample_list_kernel(Pyr aoPyr, Pyr depthPyr, Img histLen, Img normal, Img out, CamK cam, const unsigned* counts,
                                                                                       const unsigned* list, unsigned nseg)
{
    __shared__ Img aoLv[8], depthLv[8];
    {
        const unsigned t = threadIdx.x;
        if (t < 8u) aoLv[t] = aoPyr.l[t];
        else if (t < 16u) depthLv[t - 8u] = depthPyr.l[t - 8u];
        __syncthreads();
    }
    for (unsigned seg0 = blockIdx.x * 4u; seg0 < nseg; seg0 += gridDim.x * 4u)
    {
        const SegmentCursor c(counts, 0u, seg0, nseg);
        for (unsigned i = threadIdx.x; i < c.total; i += blockDim.x)
        {
            const unsigned it = c.entry(list, i);
            const int x = int(it & 0xffffu), y = int(it >> 16);
            st<ao_t>(out, x, y, ssao_resample_walk<EXACT>(x, y, ld<float>(depthPyr.l[0], x, y), ssao_resample_accum(ld<hl_t>(histLen, x, y)), aoLv, depthLv, normal, cam));
        }
    }
}

__global__ __launch_bounds__(256) void ssao_spatial_list_kernel(Img resampled, Img histLen, Img camzTex, Img normal, Img out, Img out2, CamK cam, SsaoK k, const unsigned* counts,
                                                                const unsigned* list, unsigned nseg)
{
    for (unsigned seg0 = blockIdx.x * 4u; seg0 < nseg; seg0 += gridDim.x * 4u)
    {
        const SegmentCursor c(counts, 1u, seg0, nseg);
        for (unsigned i = threadIdx.x; i < c.total; i += blockDim.x)
        {
            const unsigned it = c.entry(list, i);
            const int x = int(it & 0xffffu), y = int(it >> 16);
            const float accum  = ssao_spatial_accum(ld<hl_t>(histLen, x, y));
            const float result = ssao_spatial_filter(x, y, accum, camzTex, normal, cam, k, [&](int sx, int sy) __attribute__((always_inline)) { return ld<ao_t>(resampled, sx, sy); });
            st<ao_t>(out, x, y, result);
            if (out2.p) st<ao_t>(out2, x, y, result);
        }
    }
}

// ------------------------------------------------------------------------------------------------ launchers
static const dim3 kBlock(64, 4, 1);

mifx_status launch_ssao_prefilter_pyramid(hipStream_t s, const Pyr& p, const Pyr& camz, const CamK& cam, const mifx_ssao_attribs& a, bool depth16On) // p.l[0] = depth; fills p.l[1 ..] and camz.l[0 ..]
{
    const SsaoK k = make_k(a, false);
    bool zdone[8] = {};
    // (store windows on the produced levels: only when one launch reduces them all -- a level-by-level pass reads its source back from memory)
    const bool oneLaunch = pyramid_fusable_levels(p.l[0].w, p.l[0].h, p.levels - 1) == p.levels - 1 && p.levels >= 3;
    for (int lv = 1; lv < p.levels; ++lv)
    {
        MIFX_REQUIRE(oneLaunch || (p.l[lv].yn == 0 && camz.l[lv].yn == 0), "launch_ssao_prefilter_pyramid: row windows on levels that are reduced one by one");
        MIFX_REQUIRE(p.l[lv].y0 == camz.l[lv].y0 && p.l[lv].yn == camz.l[lv].yn, "launch_ssao_prefilter_pyramid: the depth and the camera-z level %d carry different row windows", lv);
    }
    for (int lv = 1; lv < p.levels;)
    {
        const int nl = pyramid_fusable_levels(p.l[lv - 1].w, p.l[lv - 1].h, p.levels - lv);
        if (nl >= 2)
        {
            PrefilterOp op{};
            op.src = p.l[lv - 1];
            op.zsrc = camz.l[lv - 1];
            zdone[lv - 1] = true;
            for (int j = 0; j < nl; ++j) { op.dst[j] = p.l[lv + j]; op.zdst[j] = camz.l[lv + j]; zdone[lv + j] = true; }
            op.proj  = cam.proj;
            op.q16   = depth16On ? 1 : 0;
            op.pairs = pair_aligned(op.src) && pair_aligned(op.zsrc) ? 1 : 0;
            // same expressions as in ssao_prefilter_mip_kernel, evaluated on the host in fp32
            const float effectRadius = 0.75f * k.EffectRadius * k.RadiusMultiplier;
            const float falloffRange = k.EffectFalloffRange * effectRadius;
            const float falloffFrom  = effectRadius - falloffRange;
            op.falloffMul = -1.0f / falloffRange;
            op.falloffAdd = falloffFrom / falloffRange + 1.0f;
            // (rows of level lv the launch covers: all of them, or those under the source's row window -- on a boundary of 2^nl source rows, so that a workgroup still
            //  holds the whole block of every texel it produces)
            const int srcRows = op.src.yn ? op.src.yn : op.src.h;
            MIFX_REQUIRE(op.src.yn == 0 || (lv == 1 && oneLaunch && op.src.y0 % (1 << nl) == 0 && (srcRows % (1 << nl) == 0 || op.src.y0 + srcRows == op.src.h)),
                         "launch_ssao_prefilter_pyramid: the source's row window [%d, %d) is not aligned to %d rows", op.src.y0, op.src.y0 + srcRows, 1 << nl);
            hipLaunchKernelGGL(ssao_prefilter_levels_kernel, dim3((p.l[lv].w + 15) / 16, ((srcRows + 1) / 2 + 15) / 16, 1), dim3(256, 1, 1), 0, s, op, nl);
            lv += nl;
        }
        else
        {
            hipLaunchKernelGGL(ssao_prefilter_mip_kernel, grid2d(p.l[lv], kBlock), kBlock, 0, s, p.l[lv - 1], p.l[lv], cam.proj, k, depth16On ? 1 : 0);
            ++lv;
        }
        MIFX_HIP_CHECK(hipGetLastError());
    }
    for (int lv = 0; lv < p.levels; ++lv)
        if (!zdone[lv])
        {
            hipLaunchKernelGGL(ssao_depth_to_camz_kernel, grid2d(p.l[lv], kBlock), kBlock, 0, s, p.l[lv], camz.l[lv], cam.proj);
            MIFX_HIP_CHECK(hipGetLastError());
        }
    return MIFX_OK;
}
mifx_status launch_ssao_downsample_depth(hipStream_t s, Img depth, Img out)
{
    hipLaunchKernelGGL(ssao_downsample_depth_kernel, grid2d(out, kBlock), kBlock, 0, s, depth, out);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_ssao_depth_to_camz(hipStream_t s, Img depth, Img camz, const CamK& cam)
{
    hipLaunchKernelGGL(ssao_depth_to_camz_kernel, grid2d(camz, kBlock), kBlock, 0, s, depth, camz, cam.proj);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_ssao_bilateral_upsample(hipStream_t s, Img depth, Img occlusion, Img out, const CamK& cam)
{
    hipLaunchKernelGGL(ssao_bilateral_upsample_kernel, grid2d(out, kBlock), kBlock, 0, s, depth, occlusion, out, cam);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
// `resolve` != nullptr: the fused resolve (ssao_temporal_kernel<true>); its lists take the layout of ssao_resolve_lists() for this launch's grid
mifx_status launch_ssao_temporal(hipStream_t s, Img currAO, Img prevAO, Img prevLen, Img reprojDepth, Img prevDepth, Img motion, Img outAO, Img outLen, const CamK& cur,
                                 const CamK& prev, const mifx_ssao_attribs& a, const SsaoResolve* resolve)
{
    const dim3 grid = grid2d(outAO, kBlock);
    if (resolve)
    {
        const unsigned nseg = grid.x * grid.y;
        unsigned* base = static_cast<unsigned*>(resolve->lists);
        const ResolveOut R{resolve->depth, resolve->resampled, resolve->out, resolve->out2, a.AlphaInterpolation, base, base + 2u * size_t(nseg), base + 2u * size_t(nseg) + 256u * size_t(nseg)};
        hipLaunchKernelGGL(ssao_temporal_kernel<true>, grid, kBlock, 0, s, currAO, prevAO, prevLen, reprojDepth, prevDepth, motion, outAO, outLen, cur, prev, make_k(a, false), R);
    }
    else
        hipLaunchKernelGGL(ssao_temporal_kernel<false>, grid, kBlock, 0, s, currAO, prevAO, prevLen, reprojDepth, prevDepth, motion, outAO, outLen, cur, prev, make_k(a, false), ResolveOut{});
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
// the two work-list passes behind launch_ssao_temporal(.., resolve): `rows5` = the image the temporal pass was launched on (same grid -> same segments)
mifx_status launch_ssao_resolve_lists(hipStream_t s, const Pyr& aoPyr, const Pyr& depthPyr, Img histLen, Img camz, Img normal, Img rows5, const SsaoResolve& r, const CamK& cam,
                                      const mifx_ssao_attribs& a)
{
    const dim3     grid5 = grid2d(rows5, kBlock);
    const unsigned nseg  = grid5.x * grid5.y;
    unsigned* base = static_cast<unsigned*>(r.lists);
    const unsigned *counts = base, *walk = base + 2u * size_t(nseg), *spatial = walk + 256u * size_t(nseg);
    const unsigned blocks = (nseg + 3u) / 4u < 2048u ? (nseg + 3u) / 4u : 2048u; // grid-stride over groups of four segments
    const bool exact = (int(cam.vw) % 16) == 0 && (int(cam.vh) % 16) == 0 && aoPyr.l[0].w == int(cam.vw) && aoPyr.l[0].h == int(cam.vh); // as launch_ssao_resample
    if (exact) hipLaunchKernelGGL(ssao_resample_list_kernel<true>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, s, aoPyr, depthPyr, histLen, normal, r.resampled, cam, counts, walk, nseg);
    else hipLaunchKernelGGL(ssao_resample_list_kernel<false>, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, s, aoPyr, depthPyr, histLen, normal, r.resampled, cam, counts, walk, nseg);
    MIFX_HIP_CHECK(hipGetLastError());
    hipLaunchKernelGGL(ssao_spatial_list_kernel, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, s, r.resampled, histLen, camz, normal, r.out, r.out2, cam, make_k(a, false), counts, spatial, nseg);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_ssao_convolute_pyramids(hipStream_t s, const Pyr& ao, const Pyr& depth, bool depth16On) // l[0] given; fills l[1 ..] of both
{
    for (int lv = 1; lv < ao.levels;)
    {
        const int nl = pyramid_fusable_levels(ao.l[lv - 1].w, ao.l[lv - 1].h, ao.levels - lv);
        if (nl >= 2)
        {
            ConvoluteOp op{};
            op.srcAO = ao.l[lv - 1];
            op.srcDepth = depth.l[lv - 1];
            for (int j = 0; j < nl; ++j) { op.dstAO[j] = ao.l[lv + j]; op.dstDepth[j] = depth.l[lv + j]; }
            op.q16   = depth16On ? 1 : 0;
            op.pairs = sizeof(Stored<ao_t>::value) == TexelBytes<ao_t>::value && pair_aligned(op.srcAO) && pair_aligned(op.srcDepth) ? 1 : 0; // (float AO texels only)
            MIFX_REQUIRE((ao.l[lv].y0 & ((1 << (nl - 1)) - 1)) == 0, "launch_ssao_convolute_pyramids: the row window of level %d starts at row %d, not on a row of level %d", lv, ao.l[lv].y0, lv + nl - 1);
            hipLaunchKernelGGL(ssao_convolute_levels_kernel, dim3((ao.l[lv].w + 15) / 16, (window_rows(ao.l[lv]) + 15) / 16, 1), dim3(256, 1, 1), 0, s, op, nl);
            lv += nl;
        }
        else
        {
            hipLaunchKernelGGL(ssao_convolute_mip_kernel, grid2d(ao.l[lv], kBlock), kBlock, 0, s, ao.l[lv - 1], depth.l[lv - 1], ao.l[lv], depth.l[lv], depth16On ? 1 : 0);
            ++lv;
        }
        MIFX_HIP_CHECK(hipGetLastError());
    }
    return MIFX_OK;
}
mifx_status launch_ssao_resample(hipStream_t s, const Pyr& aoPyr, const Pyr& depthPyr, Img histLen, Img normal, Img out, const CamK& cam)
{
    const bool exact = (int(cam.vw) % 16) == 0 && (int(cam.vh) % 16) == 0 && aoPyr.l[0].w == int(cam.vw) && aoPyr.l[0].h == int(cam.vh);
    if (exact) hipLaunchKernelGGL(ssao_resample_kernel<true>, tiled_grid(out), dim3(256, 1, 1), 0, s, aoPyr, depthPyr, histLen, normal, out, cam);
    else hipLaunchKernelGGL(ssao_resample_kernel<false>, tiled_grid(out), dim3(256, 1, 1), 0, s, aoPyr, depthPyr, histLen, normal, out, cam);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_ssao_spatial(hipStream_t s, Img occl, Img histLen, Img depth, Img camz, Img normal, Img out, Img historyOut, const CamK& cam, const mifx_ssao_attribs& a)
{
        hipLaunchKernelGGL(ssao_spatial_kernel, grid2d(out, kBlock), kBlock, 0, s, occl, histLen, depth, camz, normal, out, historyOut, cam, make_k(a, false));
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
