// oracle_ssao.cpp -- TEST INFRASTRUCTURE ONLY: hand-written CPU restatement of the SSAO passes A2..A8
// (Shaders/PostProcess/ScreenSpaceAmbientOcclusion/private/SSAO_*.fx).  Pinned against oracle/_ref by tests/test_oracle_vs_ref.py.
#include "oracle_kit.h"

using namespace ok;

namespace
{
struct SSAOAttribs // ScreenSpaceAmbientOcclusionStructures.fxh:64-98
{
    float EffectRadius, EffectFalloffRange, RadiusMultiplier, DepthMIPSamplingOffset, TemporalStabilityFactor, SpatialReconstructionRadius;
    int32_t ResetAccumulation;
    float AlphaInterpolation, BitmaskThickness;
    uint32_t Algorithm;
    float Padding0, Padding1;
};
static_assert(sizeof(SSAOAttribs) == 48, "ScreenSpaceAmbientOcclusionAttribs layout");
inline SSAOAttribs load_attribs(const void* p) { SSAOAttribs a; std::memcpy(&a, p, sizeof(a)); return a; }

constexpr float M_PI_F = 3.14159265358979f, M_HALF_PI_F = 1.57079632679490f;

inline float geometry_weight(f3 c, f3 t, f3 n, float norm) { return sat(1.0f - std::fabs(dot(t - c, n)) * norm); } // SSAO_Common.fxh:25-28
inline float fast_acos(float v)                                                                                       // SSAO_ComputeAmbientOcclusion.fx:47-53
{
    float a = std::fabs(v);
    float r = -0.156583f * a + M_HALF_PI_F;
    r *= std::sqrt(1.0f - a);
    return v >= 0.0f ? r : M_PI_F - r;
}
// SampleLevel with a point-clamp sampler and fractional LOD: nearest mip = floor(lod + 0.5), nearest texel
inline float sample_pyr_point(const ref_args* a, int slot, float u, float v, float mip)
{
    int l = clampi(int(std::floor(mip + 0.5f)), 0, a->in_mips[slot] - 1);
    return sample_point_clamp1(in_img(a, slot, l), u, v);
}
inline uint32_t occluded_sectors(float minH, float maxH, uint32_t bits) // :77-98
{
    minH = sat(minH); maxH = sat(maxH);
    uint32_t r = bits;
    if (maxH > minH)
    {
        const uint32_t n = 32u;
        uint32_t s = std::min(uint32_t(minH * float(n)), n - 1u), e = std::min(uint32_t(std::ceil(maxH * float(n))), n);
        if (e > s)
        {
            uint32_t ang = e - s;
            uint32_t field = ang >= 32u ? 0xFFFFFFFFu : ((1u << ang) - 1u);
            r |= field << s;
        }
    }
    return r;
}

int compute_ao(const ref_args* a, int algo) // ComputeAmbientOcclusionPS :132-236
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    const SSAOAttribs k = load_attribs(a->attribs);
    const Img normal = in_img(a, 1), noise = in_img(a, 2), out = out_img(a, 0);
    const float ivw = cam.viewport[2], ivh = cam.viewport[3], vw = cam.viewport[0], vh = cam.viewport[1];
    const float selfOcclusionOffset = a->ival[5] != 0 ? 0.005f : 0.00001f; // SSAO_OPTION_HALF_PRECISION_DEPTH :145-150
    const float uvScale = a->ival[6] != 0 ? 2.0f : 1.0f; // SSAO_OPTION_HALF_RESOLUTION: GetInvViewportSize() = 2 / viewport (:68-75); target and pyramid are half size
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const f2 uv{(float(x) + 0.5f) * (uvScale * ivw), (float(y) + 0.5f) * (uvScale * ivh)};
            const f3 posSS{uv.x, uv.y, sample_pyr_point(a, 0, uv.x, uv.y, 0.0f)};
            if (is_background(posSS.z)) continue; // discard: the target keeps its cleared value 1.0
            const int nx = clampi(int(std::floor(uv.x * float(normal.w()))), 0, normal.w() - 1), ny = clampi(int(std::floor(uv.y * float(normal.h()))), 0, normal.h() - 1);
            const f3 normalVS = mul_dir(normal.ld3(nx, ny), cam.view);
            f3 posVS = screen_xy_depth_to_view_space(posSS, cam.proj);
            posVS = posVS + normalVS * selfOcclusionOffset * posVS.z;
            const f3 viewVS = -normalize(posVS);
            const f2 xi = noise.ld2(x & 127, y & 127);
            const float effectRadius = k.EffectRadius * k.RadiusMultiplier;
            const float falloffRange = k.EffectFalloffRange * effectRadius;
            const float falloffFrom = effectRadius - falloffRange;
            const float falloffMul = -1.0f / falloffRange, falloffAdd = falloffFrom / falloffRange + 1.0f;
            float sampleRadius = 0.5f * effectRadius * cam.proj[0];
            if (cam.proj[15] == 0.0f) sampleRadius /= posVS.z;

            float visibility = 0.0f;
            for (int slice = 0; slice < 3; ++slice)
            {
                const float phi = (xi.x + float(slice) / 3.0f) * M_PI_F;
                const f2 omega{std::cos(phi), std::sin(phi)};
                const f3 sliceDir{omega.x, omega.y, 0.0f};
                const f3 ortho = sliceDir - dot(sliceDir, viewVS) * viewVS;
                const f3 axis = normalize(cross(sliceDir, viewVS));
                const f3 projN = normalVS - axis * dot(normalVS, axis);
                const float projNLen = length(projN);
                const float cosNorm = sat(dot(projN / projNLen, viewVS));
                const float n = sign(dot(ortho, projN)) * fast_acos(cosNorm);
                uint32_t occluded = 0u;
                f2 minCos{std::cos(n + M_HALF_PI_F), std::cos(n - M_HALF_PI_F)};
                f2 maxCos = minCos;
                f2 sampleDir{omega.x * 0.5f * sampleRadius, omega.y * -0.5f * sampleRadius};
                sampleDir.x *= vh * ivw;
                for (int si = 0; si < 3; ++si)
                {
                    const float noiseV = frac(xi.y + float(slice + si * 3) * 0.6180339887498948482f);
                    const float s = (float(si) + noiseV) / 3.0f;
                    const f2 off = s * s * sampleDir;
                    const f2 p0{posSS.x + off.x, posSS.y + off.y}, p1{posSS.x - off.x, posSS.y - off.y};
                    const float mip = clampf(std::log2(length(f2{off.x * vw, off.y * vh})) - k.DepthMIPSamplingOffset, 0.0f, 4.0f);
                    const f3 s0 = screen_xy_depth_to_view_space({p0.x, p0.y, sample_pyr_point(a, 0, p0.x, p0.y, mip)}, cam.proj);
                    const f3 s1 = screen_xy_depth_to_view_space({p1.x, p1.y, sample_pyr_point(a, 0, p1.x, p1.y, mip)}, cam.proj);
                    const f3 d0 = s0 - posVS, d1 = s1 - posVS;
                    if (algo == 2)
                    { // ComputeSampleOcclusion :100-119
                        const f3 thick = viewVS * k.BitmaskThickness;
                        const f2 w{sat(length(d0) * falloffMul + falloffAdd), sat(length(d1) * falloffMul + falloffAdd)};
                        f4 fb{fast_acos(dot(normalize(d0), viewVS)), fast_acos(dot(normalize(d0 - thick), viewVS)), fast_acos(dot(normalize(d1), viewVS)),
                              fast_acos(dot(normalize(d1 - thick), viewVS))};
                        const float nb = -n;
                        fb = {sat((-fb.x - nb + M_HALF_PI_F) / M_PI_F), sat((-fb.y - nb + M_HALF_PI_F) / M_PI_F), sat((fb.z - nb + M_HALF_PI_F) / M_PI_F),
                              sat((fb.w - nb + M_HALF_PI_F) / M_PI_F)};
                        if (w.x > 0.0f) occluded = occluded_sectors(fb.y, fb.x, occluded);
                        if (w.y > 0.0f) occluded = occluded_sectors(fb.z, fb.w, occluded);
                    }
                    else
                    { // ComputeSampleHorizons :121-130
                        const f2 dist{length(d0), length(d1)};
                        const f2 cosH{dot(d0 / dist.x, viewVS), dot(d1 / dist.y, viewVS)};
                        const f2 w{sat(dist.x * falloffMul + falloffAdd), sat(dist.y * falloffMul + falloffAdd)};
                        maxCos = {fmax2(maxCos.x, lerp(minCos.x, cosH.x, w.x)), fmax2(maxCos.y, lerp(minCos.y, cosH.y, w.y))};
                    }
                }
                if (algo == 2) visibility += 1.0f - float(__builtin_popcount(occluded)) / 32.0f;
                else
                {
                    const float hx = +fast_acos(maxCos.x), hy = -fast_acos(maxCos.y);
                    if (algo == 1) visibility += 0.5f * (1.0f - std::cos(hx) + (1.0f - std::cos(hy)));
                    else
                    {
                        const float h1 = hx * 2.0f, h2 = hy * 2.0f, sinN = std::sin(n);
                        visibility += projNLen * (0.25f * ((-std::cos(h1 - n) + cosNorm + h1 * sinN) + (-std::cos(h2 - n) + cosNorm + h2 * sinN)));
                    }
                }
            }
            out.st1(x, y, visibility / 3.0f);
        }
    return 0;
}

const float kPoisson[8][3] = {{-0.4706069f, -0.4427112f, +0.6461146f}, {-0.9057375f, +0.3003471f, +0.9542373f}, {-0.3487388f, +0.4037880f, +0.5335386f},
                              {+0.1023042f, +0.6439373f, +0.6520134f}, {+0.5699277f, +0.3513750f, +0.6695386f}, {+0.2939128f, -0.1131226f, +0.3149309f},
                              {+0.7836658f, -0.4208784f, +0.8895339f}, {+0.1564120f, -0.8198990f, +0.8346850f}};
} // namespace

extern "C" {

// A1 -- SSAO_ComputeDownsampledDepth.fx:8-28 (FEATURE_FLAG_HALF_RESOLUTION). in[0]: depth; out[0]: checkerboard of min / max depth of the 2x2 blocks
int oracle_ssao_downsampled_depth(const ref_args* a)
{
    const Img depth = in_img(a, 0), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float d0 = depth.ld1z(2 * x, 2 * y), d1 = depth.ld1z(2 * x, 2 * y + 1), d2 = depth.ld1z(2 * x + 1, 2 * y), d3 = depth.ld1z(2 * x + 1, 2 * y + 1);
            const float mn = fmin2(fmin2(d0, d1), fmin2(d2, d3)), mx = fmax2(fmax2(d0, d1), fmax2(d2, d3));
            out.st1(x, y, lerp(mn, mx, float(((x + y) & 1) & 1))); // ComputeCheckerboardPattern :8-11
        }
    return 0;
}

// A4 -- SSAO_ComputeBilateralUpsampling.fx:66-139 (FEATURE_FLAG_HALF_RESOLUTION). in: 0 depth (full resolution), 1 occlusion (half resolution); cam0; out[0]: full resolution
int oracle_ssao_bilateral_upsampling(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    const Img depth = in_img(a, 0), occl = in_img(a, 1), out = out_img(a, 0);
    const float vw = cam.viewport[0], vh = cam.viewport[1], ivw = cam.viewport[2], ivh = cam.viewport[3];
    const int   hw = int(0.5f * vw), hh = int(0.5f * vh); // int2(0.5 * f4ViewportSize.xy)
    auto depth_weight = [&](float center, float guide, float sigma) { // ComputeDepthWeight :66-72
        const float z0 = depth_to_camera_z(center, cam.proj), z1 = depth_to_camera_z(guide, cam.proj);
        const float alpha = std::fabs(z0 - z1) / fmax2(z0, 1e-6f);
        return std::exp(-(alpha * alpha) / (2.0f * sigma * sigma));
    };
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float center = depth.ld1z(x, y);
            if (is_background(center)) { out.st1(x, y, 1.0f); continue; }
            const int cx = int(0.5f * float(x)), cy = int(0.5f * float(y)); // int2(0.5 * floor(Position))
            float sum = 0.0f, wsum = 0.0f;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                {
                    const int lx = clampi(cx + dx, 0, hw - 1), ly = clampi(cy + dy, 0, hh - 1); // ClampScreenCoord
                    const float u = 2.0f * (float(lx) + 0.5f) * ivw, v = 2.0f * (float(ly) + 0.5f) * ivh;
                    const float signal = occl.ld1z(lx, ly);
                    const float guide  = sample_linear_clamp1(depth, u, v);
                    const float ws = spatial_weight(float(dx * dx + dy * dy), 0.9f); // SSAO_BILATERAL_UPSAMPLING_SIGMA
                    const float wz = depth_weight(center, guide, 0.0075f);           // SSAO_BILATERAL_UPSAMPLING_DEPTH_SIGMA
                    sum += ws * wz * signal;
                    wsum += ws * wz;
                }
            out.st1(x, y, wsum > 0.0f ? sum / wsum : sample_linear_clamp1(occl, 2.0f * (float(cx) + 0.5f) * ivw, 2.0f * (float(cy) + 0.5f) * ivh));
        }
    return 0;
}

// A2 -- SSAO_ComputePrefilteredDepthBuffer.fx:42-121. in[0]: previous mip; cam0; attribs; out[0]: next mip
int oracle_ssao_prefiltered_depth_mip(const ref_args* a)
{
    const Camera cam = load_camera(a->cam0);
    const SSAOAttribs k = load_attribs(a->attribs);
    const Img src = in_img(a, 0), dst = out_img(a, 0);
    const bool oddW = (src.w() & 1) != 0, oddH = (src.h() & 1) != 0;
#pragma omp parallel for
    for (int y = 0; y < dst.h(); ++y)
        for (int x = 0; x < dst.w(); ++x)
        {
            float s[9];
            int n = 0;
            auto tap = [&](int ox, int oy) { s[n++] = depth_to_camera_z(src.ld1c(2 * x + ox, 2 * y + oy), cam.proj); };
            tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
            if (oddW) { tap(2, 0); tap(2, 1); }
            if (oddH) { tap(0, 2); tap(1, 2); }
            if (oddW && oddH) tap(2, 2);
            float wd = s[0];
            for (int i = 1; i < n; ++i) wd = fmin2(wd, s[i]);
            const float effectRadius = 0.75f * k.EffectRadius * k.RadiusMultiplier;
            const float falloffRange = k.EffectFalloffRange * effectRadius;
            const float falloffFrom = effectRadius - falloffRange;
            const float falloffMul = -1.0f / falloffRange, falloffAdd = falloffFrom / falloffRange + 1.0f;
            float dsum = 0.0f, wsum = 0.0f;
            for (int i = 0; i < n; ++i)
            {
                float w = sat(std::fabs(wd - s[i]) * falloffMul + falloffAdd);
                dsum += w * s[i];
                wsum += w;
            }
            dst.st1(x, y, sat(camera_z_to_depth(dsum / wsum, cam.proj)));
        }
    return 0;
}

// A3 -- SSAO_ComputeAmbientOcclusion.fx:132-236. in: 0 depth pyramid (5 mips), 1 normal, 2 blue noise ZW; cam0; attribs; out[0]: AO (pre-filled with 1)
int oracle_ssao_compute_ao_gtao(const ref_args* a) { return compute_ao(a, 0); }
int oracle_ssao_compute_ao_hbao(const ref_args* a) { return compute_ao(a, 1); }
int oracle_ssao_compute_ao_vbao(const ref_args* a) { return compute_ao(a, 2); }

// A5 -- SSAO_ComputeTemporalAccumulation.fx:76-180
// in: 0 curr AO, 1 prev AO, 2 prev history length, 3 reprojected depth, 4 prev depth, 5 closest motion; cam0, cam1; attribs; out: 0 AO, 1 length (pre-filled with 1)
int oracle_ssao_temporal_accumulation(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cur = load_camera(a->cam0), prev = load_camera(a->cam1);
    const SSAOAttribs k = load_attribs(a->attribs);
    const Img currAO = in_img(a, 0), prevAO = in_img(a, 1), prevLen = in_img(a, 2), currDepth = in_img(a, 3), prevDepth = in_img(a, 4), motionTex = in_img(a, 5);
    const Img outAO = out_img(a, 0), outLen = out_img(a, 1);
    const float vw = cur.viewport[0], vh = cur.viewport[1];
    const int W = int(vw), H = int(vh);
#pragma omp parallel for
    for (int y = 0; y < outAO.h(); ++y)
        for (int x = 0; x < outAO.w(); ++x)
        {
            const float depth = currDepth.ld1(x, y);
            if (is_background(depth)) continue;
            const f2 m = motionTex.ld2(x, y);
            const f2 motion{m.x * 0.5f, m.y * -0.5f};
            const f2 prevLoc{(float(x) + 0.5f) - motion.x * vw, (float(y) + 0.5f) - motion.y * vh};
            const float currZ = depth_to_camera_z(depth, cur.proj);
            const Bilinear b = bilinear_uc(prevLoc.x, prevLoc.y, W, H);
            auto similar = [&](int px, int py) {
                float pz = depth_to_camera_z(prevDepth.ld1(px, py), prev.proj);
                return std::fabs(1.0f - currZ / pz) < 0.01f ? 1.0f : 0.0f;
            };
            const f4 w{b.w00 * similar(b.x0, b.y0), b.w10 * similar(b.x1, b.y0), b.w01 * similar(b.x0, b.y1), b.w11 * similar(b.x1, b.y1)};
            const float total = dot(w, splat4(1.0f));
            float occ = 1.0f, hist = 1.0f;
            const bool ok_ = total > 0.01f && !k.ResetAccumulation;
            if (ok_)
            {
                const f4 po{prevAO.ld1(b.x0, b.y0), prevAO.ld1(b.x1, b.y0), prevAO.ld1(b.x0, b.y1), prevAO.ld1(b.x1, b.y1)};
                f4 h{prevLen.ld1(b.x0, b.y0), prevLen.ld1(b.x1, b.y0), prevLen.ld1(b.x0, b.y1), prevLen.ld1(b.x1, b.y1)};
                h = min4(h + splat4(1.0f), splat4(16.0f));
                occ = dot(po, w) / total;
                hist = dot(h, w) / total;
                float m1 = 0.0f, m2 = 0.0f;
                for (int dx = -1; dx <= 1; ++dx)
                    for (int dy = -1; dy <= 1; ++dy)
                    {
                        float s = currAO.ld1(clampi(x + dx, 0, W - 1), clampi(y + dy, 0, H - 1));
                        m1 += s;
                        m2 += s * s;
                    }
                const float mean = m1 / 9.0f;
                const float var = (m2 / 9.0f) - (mean * mean);
                const float sd = std::sqrt(fmax2(var, 0.0f));
                const float aspect = vw * cur.viewport[3];
                const float mf = sat(1.025f - length(f2{motion.x * aspect, motion.y}) * 128.0f);
                const float gamma = lerp(0.5f, 2.5f, mf * mf);
                const bool inside = (mean - gamma * sd) < occ && occ < (mean + gamma * sd);
                hist = inside ? hist : fmax2(1.0f, mf * hist);
            }
            outAO.st1(x, y, lerp(occ, currAO.ld1(x, y), 1.0f / hist));
            outLen.st1(x, y, hist);
        }
    return 0;
}

// A6 -- SSAO_ComputeConvolutedDepthHistory.fx:41-110. in[0]: AO previous mip, in[1]: depth previous mip; out[0], out[1]: next mips
int oracle_ssao_convoluted_history_mip(const ref_args* a)
{
    const Img sa = in_img(a, 0), sd = in_img(a, 1), da = out_img(a, 0), dd = out_img(a, 1);
    const bool oddW = (sa.w() & 1) != 0, oddH = (sa.h() & 1) != 0;
#pragma omp parallel for
    for (int y = 0; y < da.h(); ++y)
        for (int x = 0; x < da.w(); ++x)
        {
            float av = 0.0f, dv = 0.0f;
            int n = 0;
            auto tap = [&](int ox, int oy) { av += sa.ld1c(2 * x + ox, 2 * y + oy); dv += sd.ld1c(2 * x + ox, 2 * y + oy); ++n; };
            tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
            if (oddW) { tap(2, 0); tap(2, 1); }
            if (oddH) { tap(0, 2); tap(1, 2); }
            if (oddW && oddH) tap(2, 2);
            da.st1(x, y, av / float(n));
            dd.st1(x, y, dv / float(n));
        }
    return 0;
}

// A7 -- SSAO_ComputeResampledHistory.fx:56-113. in: 0 AO pyramid, 1 depth pyramid, 2 history length, 3 normal; cam0; out[0]
int oracle_ssao_resampled_history(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    const Img histLen = in_img(a, 2), normal = in_img(a, 3), out = out_img(a, 0);
    const float vw = cam.viewport[0], vh = cam.viewport[1], ivw = cam.viewport[2], ivh = cam.viewport[3];
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float depth = in_img(a, 1, 0).ld1(x, y);
            const float hist = histLen.ld1(x, y);
            const float accum = (hist - 1.0f) / 4.0f;
            if (is_background(depth) || accum >= 1.0f) { out.st1(x, y, in_img(a, 0, 0).ld1(x, y)); continue; }
            int mip = int(4.0f * (1.0f - sat(accum)));
            const f2 pos{float(x) + 0.5f, float(y) + 0.5f};
            const f3 posVS = screen_xy_depth_to_view_space({pos.x * ivw, pos.y * ivh, depth}, cam.proj);
            const f3 normalVS = mul_dir(normal.ld3(x, y), cam.view);
            const float planeFactor = 10.0f / (1.0f + depth_to_camera_z(depth, cam.proj));
            float osum = 0.0f, wsum = 0.0f;
            while (mip >= 0 && wsum < 0.995f)
            {
                const float inv = 1.0f / float(1u << unsigned(mip));
                const f2 mipRes{vw * inv, vh * inv}, mipLoc{pos.x * inv, pos.y * inv};
                const int lx = int(mipLoc.x - 0.5f), ly = int(mipLoc.y - 0.5f);
                const float fx = frac(mipLoc.x + 0.5f), fy = frac(mipLoc.y + 0.5f);
                const float wgt[4] = {(1.0f - fx) * (1.0f - fy), fx * (1.0f - fy), (1.0f - fx) * fy, fx * fy};
                osum = 0.0f; wsum = 0.0f;
                const Img dm = in_img(a, 1, mip), am = in_img(a, 0, mip);
                for (int s = 0; s < 4; ++s)
                {
                    const int sx = lx + (s & 1), sy = ly + (s >> 1);
                    const f2 tc{(float(sx) + 0.5f) * (1.0f / mipRes.x), (float(sy) + 0.5f) * (1.0f / mipRes.y)};
                    const float sdv = sample_linear_clamp1(dm, tc.x, tc.y);
                    const float so = sample_point_clamp1(am, tc.x, tc.y);
                    const f3 sVS = screen_xy_depth_to_view_space({tc.x, tc.y, sdv}, cam.proj);
                    const float wz = geometry_weight(posVS, sVS, normalVS, planeFactor);
                    osum += so * wgt[s] * wz;
                    wsum += wgt[s] * wz;
                }
                --mip;
            }
            out.st1(x, y, osum / wsum);
        }
    return 0;
}

// A8 -- SSAO_ComputeSpatialReconstruction.fx:43-108. in: 0 resampled AO, 1 history length, 2 depth, 3 normal; cam0; attribs; out[0]
int oracle_ssao_spatial_reconstruction(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    const SSAOAttribs k = load_attribs(a->attribs);
    const Img occl = in_img(a, 0), histLen = in_img(a, 1), depthTex = in_img(a, 2), normal = in_img(a, 3), out = out_img(a, 0);
    const float ivw = cam.viewport[2], ivh = cam.viewport[3];
    const int W = int(cam.viewport[0]), H = int(cam.viewport[1]);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float hist = histLen.ld1(x, y), depth = depthTex.ld1(x, y);
            const float accum = std::pow(std::fabs((hist - 1.0f) / 8.0f), 0.2f);
            if (is_background(depth) || accum >= 1.0f) { out.st1(x, y, lerp(1.0f, occl.ld1(x, y), k.AlphaInterpolation)); continue; }
            const f2 pos{float(x) + 0.5f, float(y) + 0.5f};
            const f3 posVS = screen_xy_depth_to_view_space({pos.x * ivw, pos.y * ivh, depth}, cam.proj);
            const f3 normalVS = mul_dir(normal.ld3(x, y), cam.view);
            const float angle = 2.0f * M_PI_F * bayer4x4(uint32_t(x), uint32_t(y), cam.frameIndex);
            const f4 rot{std::cos(angle), std::sin(angle), -std::sin(angle), std::cos(angle)};
            const float radius = lerp(0.0f, k.SpatialReconstructionRadius, 1.0f - sat(accum));
            const float planeFactor = 10.0f / (1.0f + depth_to_camera_z(depth, cam.proj));
            float osum = 0.0f, wsum = 0.0f;
            for (int s = 0; s < 8; ++s)
            {
                const f2 xi = rotate_vector(rot, {kPoisson[s][0], kPoisson[s][1]});
                const int sx = clampi(int(pos.x + radius * xi.x), 0, W - 1), sy = clampi(int(pos.y + radius * xi.y), 0, H - 1);
                const f3 sVS = screen_xy_depth_to_view_space({(float(sx) + 0.5f) * ivw, (float(sy) + 0.5f) * ivh, depthTex.ld1(sx, sy)}, cam.proj);
                const float ws = spatial_weight(kPoisson[s][2] * kPoisson[s][2], 0.9f);
                const float wz = geometry_weight(posVS, sVS, normalVS, planeFactor);
                osum += ws * wz * occl.ld1(sx, sy);
                wsum += ws * wz;
            }
            const float o = wsum > 0.0f ? osum / wsum : occl.ld1(x, y);
            out.st1(x, y, lerp(1.0f, o, k.AlphaInterpolation));
        }
    return 0;
}

} // extern "C"
