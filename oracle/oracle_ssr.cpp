// oracle_ssr.cpp -- TEST INFRASTRUCTURE ONLY: hand-written CPU restatement of the SSR passes R1..R7
// (Shaders/PostProcess/ScreenSpaceReflection/private/SSR_*.fx).  Pinned against oracle/_ref by tests/test_oracle_vs_ref.py.
// Masked passes (R4-R7) skip texels whose mask is 0 (the caller pre-fills the targets with the clear value 0).
#include "oracle_kit.h"

using namespace ok;

namespace
{
struct SSRAttribs // ScreenSpaceReflectionStructures.fxh:43-80
{
    float DepthBufferThickness, RoughnessThreshold;
    uint32_t MostDetailedMip;
    int32_t IsRoughnessPerceptual;
    uint32_t RoughnessChannel, MaxTraversalIntersections;
    float GGXImportanceSampleBias, SpatialReconstructionRadius, TemporalRadianceStabilityFactor, TemporalVarianceStabilityFactor;
    float BilateralCleanupSpatialSigmaFactor, AlphaInterpolation;
};
static_assert(sizeof(SSRAttribs) == 48, "ScreenSpaceReflectionAttribs layout");
inline SSRAttribs load_attribs(const void* p) { SSRAttribs a; std::memcpy(&a, p, sizeof(a)); return a; }

constexpr float FLT_EPS_ = 5.960464478e-8f, FLT_MAX_ = 3.402823466e+38f;
constexpr int MAX_MIP = 6;
inline bool is_reflection_sample(float r, float d, float thr) { return r <= thr && !is_background(d); } // SSR_Common.fxh:57-60
inline uint32_t as_uint(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

inline float load_hiz(const ref_args* a, int slot, int x, int y, int mip) { return in_img(a, slot, mip).ld1z(x, y); }

f3 hierarchical_raymarch(const ref_args* a, int hizSlot, f3 origin, f3 dir, f2 screen, int mdm, uint32_t maxIter, bool& valid) // SSR_ComputeIntersection.fx:139-189
{
    const f3 invDir{dir.x != 0.0f ? 1.0f / dir.x : FLT_MAX_, dir.y != 0.0f ? 1.0f / dir.y : FLT_MAX_, dir.z != 0.0f ? 1.0f / dir.z : FLT_MAX_};
    int curMip = mdm;
    f2 mipRes = screen * (1.0f / float(1 << curMip));
    f2 invMipRes{1.0f / mipRes.x, 1.0f / mipRes.y};
    f2 uvOff = 0.005f * float(1 << mdm) / screen;
    uvOff.x = dir.x < 0.0f ? -uvOff.x : uvOff.x;
    uvOff.y = dir.y < 0.0f ? -uvOff.y : uvOff.y;
    const f2 floorOff{dir.x < 0.0f ? 0.0f : 1.0f, dir.y < 0.0f ? 0.0f : 1.0f};
    float curT;
    f3 pos;
    { // InitialAdvanceRay :66-86
        const f2 mp = mipRes * f2{origin.x, origin.y};
        f2 plane{std::floor(mp.x) + floorOff.x, std::floor(mp.y) + floorOff.y};
        plane = plane * invMipRes + uvOff;
        const f2 t{plane.x * invDir.x - origin.x * invDir.x, plane.y * invDir.y - origin.y * invDir.y};
        curT = fmin2(t.x, t.y);
        pos = origin + curT * dir;
    }
    uint32_t idx = 0u;
    while (idx < maxIter && curMip >= mdm)
    {
        const f2 mp = mipRes * f2{pos.x, pos.y};
        const float surf = load_hiz(a, hizSlot, int(mp.x), int(mp.y), curMip);
        f2 plane{std::floor(mp.x) + floorOff.x, std::floor(mp.y) + floorOff.y}; // AdvanceRay :88-137
        plane = plane * invMipRes + uvOff;
        f3 t{plane.x * invDir.x - origin.x * invDir.x, plane.y * invDir.y - origin.y * invDir.y, surf * invDir.z - origin.z * invDir.z};
        t.z = (g_reversed_depth ? dir.z < 0.0f : dir.z > 0.0f) ? t.z : FLT_MAX_; // :108-113
        const float tmin = fmin2(fmin2(t.x, t.y), t.z);
        const bool above = g_reversed_depth ? surf < pos.z : surf > pos.z;        // :118-124
        const bool skipped = as_uint(tmin) != as_uint(t.z) && above;
        curT = above ? tmin : curT;
        pos = origin + curT * dir;
        const bool nextOut = skipped && (curMip >= MAX_MIP);
        if (!nextOut)
        {
            curMip += skipped ? 1 : -1;
            mipRes = mipRes * (skipped ? 0.5f : 2.0f);
            invMipRes = invMipRes * (skipped ? 2.0f : 0.5f);
        }
        ++idx;
    }
    valid = idx <= maxIter;
    return pos;
}
float edge_vignette(f2 hit, f2 screen) // :191-196
{
    const f2 fov{0.05f * (screen.y / screen.x), 0.05f * 1.0f};
    return (smoothstep(0.0f, fov.x, hit.x) * (1.0f - smoothstep(1.0f - fov.x, 1.0f, hit.x))) * (smoothstep(0.0f, fov.y, hit.y) * (1.0f - smoothstep(1.0f - fov.y, 1.0f, hit.y)));
}
// hitPrev: the hit moved back along its motion vector (SSR_OPTION_PREVIOUS_FRAME, :230-231); pass the hit itself otherwise
float validate_hit(const ref_args* a, int hizSlot, const Img& normal, f3 hit, f2 hitPrev, bool previousFrame, f2 uv, f3 rayWS, f2 screen, float thickness, const float* proj) // :199-252
{
    if (hit.x < 0.0f || hit.y < 0.0f || hit.x > 1.0f || hit.y > 1.0f) return 0.0f;
    if (std::fabs(hit.x - uv.x) < (2.0f / screen.x) && std::fabs(hit.y - uv.y) < (2.0f / screen.y)) return 0.0f;
    const int tx = int(screen.x * hit.x), ty = int(screen.y * hit.y);
    const float surf = load_hiz(a, hizSlot, tx, ty, 0);
    if (is_background(surf)) return 0.0f;
    const f3 hn = normal.inside(tx, ty) ? normal.ld3(tx, ty) : f3{0.f, 0.f, 0.f};
    if (dot(hn, rayWS) > 0.0f) return 0.0f;
    const f3 sVS = screen_xy_depth_to_view_space({hit.x, hit.y, surf}, proj), hVS = screen_xy_depth_to_view_space(hit, proj);
    const float dist = length(sVS - hVS);
    float conf = 1.0f - smoothstep(0.0f, thickness, dist * (1.0f / (sVS.z + FLT_EPS_)));
    conf *= conf;
    const float vignette = previousFrame ? fmin2(edge_vignette(hitPrev, screen), edge_vignette({hit.x, hit.y}, screen)) : edge_vignette({hit.x, hit.y}, screen);
    return vignette * conf;
}
inline float disocclusion(float a, float b) // SSR_ComputeTemporalAccumulation.fx:113-118
{
    a = std::fabs(a); b = std::fabs(b);
    return std::exp(-std::fabs(a - b) / fmax2(fmax2(a, b), 1e-6f));
}
const float kPoisson[8][3] = {{-0.4706069f, -0.4427112f, +0.6461146f}, {-0.9057375f, +0.3003471f, +0.9542373f}, {-0.3487388f, +0.4037880f, +0.5335386f},
                              {+0.1023042f, +0.6439373f, +0.6520134f}, {+0.5699277f, +0.3513750f, +0.6695386f}, {+0.2939128f, -0.1131226f, +0.3149309f},
                              {+0.7836658f, -0.4208784f, +0.8895339f}, {+0.1564120f, -0.8198990f, +0.8346850f}};
} // namespace

extern "C" {

// R1 -- SSR_ComputeHierarchicalDepthBuffer.fx:24-71. in[0]: previous mip; out[0]: next mip
int oracle_ssr_hiz_mip(const ref_args* a)
{
    set_depth_convention(a);
    const Img src = in_img(a, 0), dst = out_img(a, 0);
    const bool oddW = (src.w() & 1) != 0, oddH = (src.h() & 1) != 0;
#pragma omp parallel for
    for (int y = 0; y < dst.h(); ++y)
        for (int x = 0; x < dst.w(); ++x)
        {
            float m = depth_far_plane();
            auto tap = [&](int ox, int oy) { m = closest_depth(m, src.ld1c(2 * x + ox, 2 * y + oy)); };
            tap(0, 0); tap(0, 1); tap(1, 0); tap(1, 1);
            if (oddW) { tap(2, 0); tap(2, 1); }
            if (oddH) { tap(0, 2); tap(1, 2); }
            if (oddW && oddH) tap(2, 2);
            dst.st1(x, y, m);
        }
    return 0;
}

// R2 -- SSR_ComputeStencilMaskAndExtractRoughness.fx:13-40. in[0]: material (c=4), in[1]: depth; attribs; out[0]: roughness (every texel), out[1]: mask
int oracle_ssr_mask_roughness(const ref_args* a)
{
    set_depth_convention(a);
    const SSRAttribs k = load_attribs(a->attribs);
    const Img mat = in_img(a, 0), depth = in_img(a, 1), ro = out_img(a, 0), mo = out_img(a, 1);
#pragma omp parallel for
    for (int y = 0; y < ro.h(); ++y)
        for (int x = 0; x < ro.w(); ++x)
        {
            const f4 m = mat.ld4(x, y);
            const f4 sel{k.RoughnessChannel == 0u ? 1.0f : 0.0f, k.RoughnessChannel == 1u ? 1.0f : 0.0f, k.RoughnessChannel == 2u ? 1.0f : 0.0f,
                         k.RoughnessChannel == 3u ? 1.0f : 0.0f};
            float r = dot(m, sel);
            if (!k.IsRoughnessPerceptual) r = std::sqrt(r);
            ro.st1(x, y, r);
            mo.st1(x, y, is_reflection_sample(r, depth.ld1(x, y), k.RoughnessThreshold) ? 1.0f : 0.0f);
        }
    return 0;
}

// R3 -- SSR_ComputeDownsampledStencilMask.fx:13-61 (FEATURE_FLAG_HALF_RESOLUTION). in: 0 roughness, 1 depth; attribs; out[0]: half-resolution mask
int oracle_ssr_downsampled_mask(const ref_args* a)
{
    set_depth_convention(a);
    const SSRAttribs k = load_attribs(a->attribs);
    const Img rough = in_img(a, 0), depth = in_img(a, 1), out = out_img(a, 0);
    const bool oddW = (depth.w() & 1) != 0, oddH = (depth.h() & 1) != 0;
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            float minDepth = depth_far_plane(), maxRough = 0.0f;
            auto tap = [&](int ox, int oy) {
                const int lx = clampi(2 * x + ox, 0, depth.w() - 1), ly = clampi(2 * y + oy, 0, depth.h() - 1); // ClampScreenCoord to the depth texture's size
                minDepth = closest_depth(minDepth, depth.ld1(lx, ly));
                maxRough = fmax2(maxRough, rough.ld1z(lx, ly));
            };
            tap(0, 0); tap(1, 0); tap(0, 1); tap(1, 1);
            if (oddW) { tap(2, 0); tap(2, 1); }
            if (oddH) { tap(0, 2); tap(1, 2); }
            if (oddW && oddH) tap(2, 2);
            out.st1(x, y, is_reflection_sample(maxRough, minDepth, k.RoughnessThreshold) ? 1.0f : 0.0f);
        }
    return 0;
}

// R4 -- SSR_ComputeIntersection.fx:254-335. in: 0 radiance, 1 normal, 2 roughness, 3 blue noise XY, 4 Hi-Z (7 mips), 5 mask, 6 motion (ival[0] != 0: SSR_OPTION_PREVIOUS_FRAME);
// cam0; attribs; out: 0 specular, 1 dir*len+pdf
int oracle_ssr_intersection(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    const SSRAttribs k = load_attribs(a->attribs);
    const Img radiance = in_img(a, 0), normal = in_img(a, 1), roughTex = in_img(a, 2), noise = in_img(a, 3), mask = in_img(a, 5), o0 = out_img(a, 0), o1 = out_img(a, 1);
    const f2 screen{cam.viewport[0], cam.viewport[1]};
    const bool previousFrame = a->ival[0] != 0;
    const bool halfRes = a->ival[6] != 0; // SSR_OPTION_HALF_RESOLUTION: targets, mask and noise are indexed by the half-resolution texel (tx, ty)
#pragma omp parallel for schedule(dynamic, 2)
    for (int ty = 0; ty < o0.h(); ++ty)
        for (int tx = 0; tx < o0.w(); ++tx)
        {
            if (mask.ld1(tx, ty) == 0.0f) continue;
            int x = tx, y = ty; // the full-resolution pixel whose ray is traced (:283-288)
            if (halfRes)
            {
                const uint32_t idx = ((uint32_t(tx) & 3u) << 3u) + ((uint32_t(ty) & 3u) << 1u);
                const uint32_t sampleIdx = (1320229860u >> idx) & 3u; // ComputeHalfResolutionOffset, PostFX_Common.fxh:45-55
                x = 2 * tx + int(sampleIdx & 1u);
                y = 2 * ty + int(sampleIdx >> 1u);
            }
            const f2 uv{(float(x) + 0.5f) * cam.viewport[2], (float(y) + 0.5f) * cam.viewport[3]};
            const f3 normalVS = mul_dir(normal.inside(x, y) ? normal.ld3(x, y) : f3{0.f, 0.f, 0.f}, cam.view);
            const float rough = roughTex.ld1z(x, y);
            const int mdm = rough < 0.01f ? 0 : int(k.MostDetailedMip);
            const f2 mipRes = screen * (1.0f / float(1 << mdm));
            const f3 originSS{uv.x, uv.y, load_hiz(a, 4, int(uv.x * mipRes.x), int(uv.y * mipRes.y), mdm)};
            const f3 originVS = screen_xy_depth_to_view_space(originSS, cam.proj);
            const f3 view = -normalize(originVS);
            // SampleReflectionVector :254-278
            const float alpha = rough * rough;
            const f3 N = normalVS;
            const f3 T = normalize(cross(N, std::fabs(N.y) > 0.5f ? f3{1.f, 0.f, 0.f} : f3{0.f, 1.f, 0.f}));
            const f3 B = cross(T, N);
            f2 xi = noise.ld2(tx & 127, ty & 127); // LoadRandomVector2D(int2(VSOut.f4PixelPos.xy))
            xi.y = lerp(xi.y, 0.0f, k.GGXImportanceSampleBias);
            const f3 viewTS{dot(T, view), dot(B, view), dot(N, view)};
            const f3 micro = smith_ggx_sample_visible_normal_sc(viewTS, alpha, alpha, xi.x, xi.y);
            const f3 sampTS = reflect(-viewTS, micro);
            const float pdf = smith_ggx_masking(viewTS.z, alpha) * normal_distribution_ggx(micro.z, alpha) / (4.0f * viewTS.z + FLT_EPS_);
            const f3 dirVS = sampTS.x * T + sampTS.y * B + sampTS.z * N;
            const f3 dirSS = project_position(originVS + dirVS, cam.proj) - originSS;
            const f3 dirWS = mul_dir(dirVS, cam.viewInv);
            bool valid = false;
            const f3 hitSS = hierarchical_raymarch(a, 4, originSS, dirSS, screen, mdm, k.MaxTraversalIntersections, valid);
            const f3 hitVS = screen_xy_depth_to_view_space(hitSS, cam.proj);
            f2 hitPrev{hitSS.x, hitSS.y};
            if (previousFrame) // :310-312
            {
                const f2 motion = in_img(a, 6).ld2z(int(screen.x * hitSS.x), int(screen.y * hitSS.y)) * f2{0.5f, -0.5f};
                hitPrev = f2{hitSS.x, hitSS.y} - motion;
            }
            const float conf = valid ? validate_hit(a, 4, normal, hitSS, hitPrev, previousFrame, uv, dirWS, screen, k.DepthBufferThickness, cam.proj) : 0.0f;
            f3 refl{0.f, 0.f, 0.f};
            if (conf > 0.0f)
            {
                const int rx = int(screen.x * hitPrev.x), ry = int(screen.y * hitPrev.y);
                if (radiance.inside(rx, ry)) refl = radiance.ld3(rx, ry);
            }
            o0.st4(tx, ty, mk4(refl, conf));
            o1.st4(tx, ty, mk4(dirWS * length(hitVS - originVS), pdf));
        }
    return 0;
}

// R5 -- SSR_ComputeSpatialReconstruction.fx:60-175. in: 0 roughness, 1 normal, 2 depth, 3 dir+pdf, 4 specular, 5 mask; cam0; attribs; out: 0 radiance, 1 variance, 2 depth
int oracle_ssr_spatial_reconstruction(const ref_args* a)
{
    const bool halfRes = a->ival[6] != 0; // SSR_OPTION_HALF_RESOLUTION: the ray textures (in 3, 4) are half size
    const Camera cam = load_camera(a->cam0);
    const SSRAttribs k = load_attribs(a->attribs);
    const Img roughTex = in_img(a, 0), normal = in_img(a, 1), depthTex = in_img(a, 2), dirPdf = in_img(a, 3), spec = in_img(a, 4), mask = in_img(a, 5);
    const Img o0 = out_img(a, 0), o1 = out_img(a, 1), o2 = out_img(a, 2);
    const int W = int(cam.viewport[0]), H = int(cam.viewport[1]);
    const f3 camPos{cam.pos[0], cam.pos[1], cam.pos[2]};
#pragma omp parallel for
    for (int y = 0; y < o0.h(); ++y)
        for (int x = 0; x < o0.w(); ++x)
        {
            if (mask.ld1(x, y) == 0.0f) continue;
            const f2 pos{float(x) + 0.5f, float(y) + 0.5f};
            const f3 posWS = inv_project_position({pos.x * cam.viewport[2], pos.y * cam.viewport[3], depthTex.ld1(x, y)}, cam.viewProjInv);
            const f3 N = normal.ld3(x, y);
            const f3 V = normalize(camPos - posWS);
            const float NdotV = sat(dot(N, V));
            const float rough = roughTex.ld1(x, y);
            const float radius = lerp(0.0f, k.SpatialReconstructionRadius, sat(5.0f * rough));
            const float angle = 2.0f * 3.14159265358979f * bayer4x4(uint32_t(x), uint32_t(y), cam.frameIndex);
            const f4 rot{std::cos(angle), std::sin(angle), -std::sin(angle), std::cos(angle)};
            f4 colorSum = splat4(0.0f);
            float wsum = 0.0f, variance = 0.0f, mean = 0.0f, nearest = 0.0f;
            for (int s = 0; s < 8; ++s)
            {
                const f2 xi = rotate_vector(rot, {kPoisson[s][0], kPoisson[s][1]});
                int sx, sy;
                if (halfRes) // :153-154
                {
                    sx = clampi(int(0.5f * (std::floor(pos.x) + radius * xi.x) + 0.5f), 0, int(0.5f * cam.viewport[0]) - 1);
                    sy = clampi(int(0.5f * (std::floor(pos.y) + radius * xi.y) + 0.5f), 0, int(0.5f * cam.viewport[1]) - 1);
                }
                else
                {
                    sx = clampi(int(pos.x + radius * xi.x), 0, W - 1);
                    sy = clampi(int(pos.y + radius * xi.y), 0, H - 1);
                }
                const float ws = spatial_weight(kPoisson[s][2] * kPoisson[s][2], 0.9f);
                float wgt, rayLen; // ComputeWeightRayLength :60-88
                const f4 dp = dirPdf.ld4(sx, sy);
                const float len = length(xyz(dp));
                if (len < 1e-6f) { wgt = 1e-6f; rayLen = 1e-6f; }
                else
                {
                    const f3 L = xyz(dp) / len;
                    const float alpha = rough * rough;
                    const f3 Hh = normalize(L + V);
                    const float NdotH = sat(dot(N, Hh)), NdotL = sat(dot(N, L));
                    float brdf = smith_ggx_visibility_correlated(NdotL, NdotV, alpha) * normal_distribution_ggx(NdotH, alpha) * NdotL;
                    brdf *= ws;
                    wgt = fmax2(brdf / fmax2(dp.w, 1e-5f), 1e-6f);
                    rayLen = len;
                }
                const f4 c = spec.ld4(sx, sy);
                colorSum = colorSum + wgt * c; // ComputeWeightedVariance :90-100
                wsum += wgt;
                const float value = luminance601(xyz(c));
                const float prevMean = mean;
                mean += wgt * (1.0f / wsum) * (value - prevMean);
                variance += wgt * (value - prevMean) * (value - mean);
                if (wgt > 1.0e-6f) nearest = fmax2(rayLen, nearest);
            }
            o0.st4(x, y, colorSum / fmax2(wsum, 1e-6f));
            o1.st1(x, y, variance / fmax2(wsum, 1e-6f));
            o2.st1(x, y, camera_z_to_depth(length(camPos - posWS) + nearest, cam.proj)); // ComputeResolvedDepth :102-106
        }
    return 0;
}

// R6 -- SSR_ComputeTemporalAccumulation.fx:104-275.
// in: 0 motion, 1 hit depth, 2 reprojected depth, 3 curr radiance, 4 curr variance, 5 prev depth, 6 prev radiance, 7 prev variance, 8 mask; cam0, cam1; attribs
int oracle_ssr_temporal_accumulation(const ref_args* a)
{
    const Camera cur = load_camera(a->cam0), prev = load_camera(a->cam1);
    const SSRAttribs k = load_attribs(a->attribs);
    const Img motionTex = in_img(a, 0), hitDepth = in_img(a, 1), currDepth = in_img(a, 2), currRad = in_img(a, 3), currVar = in_img(a, 4), prevDepth = in_img(a, 5),
              prevRad = in_img(a, 6), prevVar = in_img(a, 7), mask = in_img(a, 8), o0 = out_img(a, 0), o1 = out_img(a, 1);
    const float vw = cur.viewport[0], vh = cur.viewport[1], ivw = cur.viewport[2], ivh = cur.viewport[3];
    const int W = int(vw), H = int(vh);
#pragma omp parallel for
    for (int y = 0; y < o0.h(); ++y)
        for (int x = 0; x < o0.w(); ++x)
        {
            if (mask.ld1(x, y) == 0.0f) continue;
            const f2 pos{float(x) + 0.5f, float(y) + 0.5f};
            f4 m1 = splat4(0.0f), m2 = splat4(0.0f); // ComputePixelStatistic :122-145
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                {
                    const f4 c = currRad.ld4(clampi(x + dx, 0, W - 1), clampi(y + dy, 0, H - 1));
                    m1 += c;
                    m2 += c * c;
                }
            const f4 mean = m1 / 9.0f;
            const f4 sd = sqrt4(max4((m2 / 9.0f) - (mean * mean), 0.0f));
            const float depth = currDepth.ld1(x, y), hd = hitDepth.ld1(x, y);
            const f2 mraw = motionTex.ld2(x, y);
            const f2 motion{mraw.x * 0.5f, mraw.y * -0.5f};
            const f2 prevInc{pos.x - motion.x * vw, pos.y - motion.y * vh};
            f2 prevHit; // ComputeReflectionHitPosition :104-110
            {
                const f2 tc{(float(x) + 0.5f) * ivw + 0.5f * cur.jitter[0], (float(y) + 0.5f) * ivh + -0.5f * cur.jitter[1]};
                const f3 pw = inv_project_position({tc.x, tc.y, hd}, cur.viewProjInv);
                const f3 pc = project_position(pw, prev.viewProj);
                prevHit = {(pc.x - 0.5f * prev.jitter[0]) * vw, (pc.y - -0.5f * prev.jitter[1]) * vh};
            }
            auto sample_prev = [&](f2 p) { return sample_linear_clamp4(prevRad, p.x * ivw, p.y * ivh); };
            const f4 cInc = sample_prev(prevInc), cHit = sample_prev(prevHit);
            const float meanLum = luminance601(xyz(mean));
            const float dInc = std::fabs(luminance601(xyz(cInc)) - meanLum), dHit = std::fabs(luminance601(xyz(cHit)) - meanLum);
            const f2 prevCoord = dInc < dHit ? prevInc : prevHit;
            // ComputeReprojection :147-222
            const float currZ = depth_to_camera_z(depth, cur.proj);
            f2 rCoord = prevCoord;
            f4 rColor = sample_prev(prevCoord);
            bool success = disocclusion(currZ, depth_to_camera_z(prevDepth.ld1z(int(prevCoord.x), int(prevCoord.y)), prev.proj)) > 0.9f;
            if (!success)
            {
                f4 bestW = splat4(0.0f);
                int bx0 = 0, by0 = 0, bx1 = 0, by1 = 0;
                float best = 0.0f;
                bool done = false;
                for (int dy = -1; dy <= 1 && !done; ++dy)
                {
                    for (int dx = -1; dx <= 1; ++dx)
                    {
                        const f2 loc{prevCoord.x + float(dx), prevCoord.y + float(dy)};
                        const Bilinear b = bilinear_uc(loc.x, loc.y, currDepth.w(), currDepth.h());
                        auto okz = [&](int px, int py) { return disocclusion(currZ, depth_to_camera_z(prevDepth.ld1(px, py), prev.proj)) > (0.9f / 2.0f) ? 1.0f : 0.0f; };
                        const f4 w{b.w00 * okz(b.x0, b.y0), b.w10 * okz(b.x1, b.y0), b.w01 * okz(b.x0, b.y1), b.w11 * okz(b.x1, b.y1)};
                        const float total = dot(w, splat4(1.0f));
                        if (total > best)
                        {
                            best = total; bestW = w; bx0 = b.x0; by0 = b.y0; bx1 = b.x1; by1 = b.y1;
                            rCoord = loc;
                            if (best > 0.9f) break;
                        }
                    }
                    if (best > 0.9f) done = true;
                }
                success = best > 0.1f;
                if (success)
                    rColor = (prevRad.ld4(bx0, by0) * bestW.x + prevRad.ld4(bx1, by0) * bestW.y + prevRad.ld4(bx0, by1) * bestW.z + prevRad.ld4(bx1, by1) * bestW.w) / best;
            }
            success = success && (rCoord.x >= 0.0f && rCoord.y >= 0.0f && rCoord.x < vw && rCoord.y < vh);
            if (success)
            {
                const f4 pr = min4(max4(rColor, mean - 2.5f * sd), mean + 2.5f * sd);
                const float pv = sample_linear_clamp1(prevVar, rCoord.x * ivw, rCoord.y * ivh);
                o0.st4(x, y, lerp(currRad.ld4(x, y), pr, k.TemporalRadianceStabilityFactor));
                o1.st1(x, y, lerp(currVar.ld1(x, y), pv, k.TemporalVarianceStabilityFactor));
            }
            else
            {
                o0.st4(x, y, currRad.ld4(x, y));
                o1.st1(x, y, 1.0f);
            }
        }
    return 0;
}

// R7 -- SSR_ComputeBilateralCleanup.fx:49-103. in: 0 depth, 1 normal, 2 roughness, 3 radiance history, 4 variance history, 5 mask; cam0; attribs; out[0]
// ddx/ddy(CameraZ): fine derivatives inside the 2x2 pixel quad (right - left, bottom - top), quad lanes outside the image replicate the nearest pixel.
int oracle_ssr_bilateral_cleanup(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    const SSRAttribs k = load_attribs(a->attribs);
    const Img depthTex = in_img(a, 0), normal = in_img(a, 1), roughTex = in_img(a, 2), radTex = in_img(a, 3), varTex = in_img(a, 4), mask = in_img(a, 5), out = out_img(a, 0);
    const int W = int(cam.viewport[0]), H = int(cam.viewport[1]);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            if (mask.ld1(x, y) == 0.0f) continue;
            const float rough = roughTex.ld1(x, y), var = varTex.ld1(x, y);
            const f3 N = normal.ld3(x, y);
            const float camZ = depth_to_camera_z(depthTex.ld1(x, y), cam.proj);
            auto cz = [&](int px, int py) { return depth_to_camera_z(depthTex.ld1(std::min(px, W - 1), std::min(py, H - 1)), cam.proj); };
            const int qx = x & ~1, qy = y & ~1;
            const f2 grad{cz(qx + 1, y) - cz(qx, y), cz(x, qy + 1) - cz(x, qy)};
            const float radius = lerp(0.0f, var > 0.001f ? 2.0f : 0.0f, sat(8.0f * rough));
            const float sigma = k.BilateralCleanupSpatialSigmaFactor;
            const int er = int(fmin2(2.0f * sigma, radius));
            f4 result = radTex.ld4(x, y);
            if (var > 0.00005f && er > 0)
            {
                f4 csum = splat4(0.0f);
                float wsum = 0.0f;
                for (int dx = -er; dx <= er; ++dx)
                    for (int dy = -er; dy <= er; ++dy)
                    {
                        const int sx = clampi(x + dx, 0, W - 1), sy = clampi(y + dy, 0, H - 1);
                        const float sd = depthTex.ld1(sx, sy), sr = roughTex.ld1(sx, sy);
                        if (!is_reflection_sample(sr, sd, k.RoughnessThreshold)) continue;
                        const float sz = depth_to_camera_z(sd, cam.proj);
                        const f2 o{float(dx), float(dy)};
                        const float ws = std::exp(-0.5f * dot(o, o) / (sigma * sigma));
                        const float wz = std::exp(-std::fabs(camZ - sz) / (1.0f * (std::fabs(dot(o, grad)) + 1e-6f)));
                        const float wn = std::pow(fmax2(0.0f, dot(N, normal.ld3(sx, sy))), 128.0f);
                        const float w = ws * wn * wz;
                        wsum += w;
                        csum += w * radTex.ld4(sx, sy);
                    }
                result = csum / fmax2(wsum, 1.0e-6f);
            }
            out.st4(x, y, {result.x, result.y, result.z, result.w * k.AlphaInterpolation});
        }
    return 0;
}

} // extern "C"
