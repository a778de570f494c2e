// oracle_shade_post.cpp -- TEST INFRASTRUCTURE ONLY: hand-written CPU restatement of
//   T1 TAA                      Shaders/PostProcess/TemporalAntiAliasing/private/TAA_ComputeTemporalAccumulation.fx
//   B1-B3 Bloom                 Shaders/PostProcess/Bloom/private/Bloom_*.fx
//   P1-P9 PBR shade             Shaders/PBR/public/PBR_Shading.fxh + Shaders/PBR/private/RenderPBR.psh (lighting half)
//   M1 composite                Hydrogent/shaders/HnPostProcess.psh:145-185
//   I1-I3 IBL precompute        Shaders/PBR/private/{PrecomputeBRDF,PrefilterEnvMap,ComputeIrradianceMap}.psh
// Pinned against oracle/_ref by tests/test_oracle_vs_ref.py.
#include "oracle_kit.h"

using namespace ok;

namespace
{
// ================================================================================================ TAA
struct TAAAttribs { float TemporalStabilityFactor; int32_t ResetAccumulation, SkipRejection; float Padding0; }; // TemporalAntiAliasingStructures.fxh:35-46

template <bool Y> inline f3 rgb_to_ycocg(f3 c) // :34-49
{
    if (!Y) return c;
    float co = c.x - c.z, t = c.z + 0.5f * co, cg = c.y - t, yy = t + 0.5f * cg;
    return {yy, co, cg};
}
template <bool Y> inline f3 ycocg_to_rgb(f3 c) // :51-66
{
    if (!Y) return c;
    float t = c.x - 0.5f * c.z, g = c.z + t, b = t - 0.5f * c.y, r = b + c.y;
    return {r, g, b};
}
inline f3 hdr_to_sdr(f3 c) { return c * (splat3(1.0f) / (splat3(1.0f) + c)); }                        // :68-71
inline f3 sdr_to_hdr(f3 c) { return c * (splat3(1.0f) / (splat3(1.0f) - c + splat3(5.960464478e-8f))); } // :73-76

template <bool GAUSS, bool BICUBIC, bool YCOCG> int taa(const ref_args* a) // ComputeTemporalAccumulationPS :229-262
{
    const Camera cur = load_camera(a->cam0), prev = load_camera(a->cam1);
    TAAAttribs k;
    std::memcpy(&k, a->attribs, sizeof(k));
    const Img currColor = in_img(a, 0), prevColor = in_img(a, 1), motionTex = in_img(a, 2), currDepth = in_img(a, 3), prevDepth = in_img(a, 4), out = out_img(a, 0);
    const float vw = cur.viewport[0], vh = cur.viewport[1], ivw = cur.viewport[2], ivh = cur.viewport[3];
    const int W = int(vw), H = int(vh);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            auto sample_curr = [&](int px, int py) { return max3(currColor.ld3(px, py), 0.0f); };
            const f2 pos{float(x) + 0.5f, float(y) + 0.5f};
            const f2 m = motionTex.ld2(x, y);
            const f2 motion{m.x * 0.5f, m.y * -0.5f};
            const f2 prevPos{pos.x - motion.x * vw, pos.y - motion.y * vh};
            const bool inside = prevPos.x >= 0.0f && prevPos.y >= 0.0f && prevPos.x < vw && prevPos.y < vh;
            if (!inside || k.ResetAccumulation) { out.st4(x, y, mk4(sample_curr(x, y), 0.5f)); continue; }
            const float aspect = vw * ivh;
            const float motionFactor = sat(1.0f - length(f2{motion.x * aspect, motion.y}) * 256.0f);
            float dis = 0.0f; // ComputeDepthDisocclusion :117-136
            {
                const int pxi = int(prevPos.x), pyi = int(prevPos.y);
                const float lc = std::fabs(depth_to_camera_z(currDepth.ld1(x, y), cur.proj));
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx)
                    {
                        const float lp = std::fabs(depth_to_camera_z(prevDepth.ld1z(pxi + dx, pyi + dy), prev.proj));
                        dis = fmax2(dis, std::exp(-std::fabs(lc - lp) / fmax2(fmax2(lc, lp), 1e-6f)));
                    }
            }
            const float depthFactor = dis > 0.9f ? 1.0f : 0.0f;
            const f3 currRGB = sample_curr(x, y);
            f4 prevRGBA;
            if (BICUBIC)
            { // SamplePrevColorCatmullRom :138-173
                const f2 texel{ivw, ivh};
                const f2 centre{std::floor(prevPos.x - 0.5f) + 0.5f, std::floor(prevPos.y - 0.5f) + 0.5f};
                const f2 f = prevPos - centre, f2_ = f * f, f3_ = f2_ * f;
                const f2 w0 = -0.5f * f3_ + f2_ - 0.5f * f;
                const f2 w1 = 1.5f * f3_ - 2.5f * f2_ + 1.0f;
                const f2 w2 = -1.5f * f3_ + 2.0f * f2_ + 0.5f * f;
                const f2 w3 = 0.5f * f3_ - 0.5f * f2_;
                const f2 w12 = w1 + w2;
                const f2 tp0 = (centre - 1.0f) * texel, tp3 = (centre + 2.0f) * texel, tp12 = (centre + w2 / w12) * texel;
                const float p0 = w12.x * w0.y, p1 = w0.x * w12.y, p2 = w12.x * w12.y, p3 = w3.x * w12.y, p4 = w12.x * w3.y;
                f4 r = splat4(0.0f);
                r += sample_linear_clamp4(prevColor, tp12.x, tp0.y) * p0;
                r += sample_linear_clamp4(prevColor, tp0.x, tp12.y) * p1;
                r += sample_linear_clamp4(prevColor, tp12.x, tp12.y) * p2;
                r += sample_linear_clamp4(prevColor, tp3.x, tp12.y) * p3;
                r += sample_linear_clamp4(prevColor, tp12.x, tp3.y) * p4;
                prevRGBA = max4(r * (1.0f / (p0 + p1 + p2 + p3 + p4)), 0.0f);
            }
            else
                prevRGBA = max4(sample_linear_clamp4(prevColor, prevPos.x * ivw, prevPos.y * ivh), 0.0f);
            const f3 currY = rgb_to_ycocg<YCOCG>(hdr_to_sdr(currRGB)), prevY = rgb_to_ycocg<YCOCG>(hdr_to_sdr(xyz(prevRGBA)));
            auto corrected = [&](float al) { return fmin2(k.TemporalStabilityFactor, sat(1.0f / (2.0f - al))); };
            if (k.SkipRejection)
            {
                out.st4(x, y, mk4(sdr_to_hdr(ycocg_to_rgb<YCOCG>(lerp(currY, prevY, prevRGBA.w))), corrected(prevRGBA.w)));
                continue;
            }
            const float gamma = lerp(0.75f, 2.5f, motionFactor * motionFactor);
            float wsum = 0.0f; // ComputePixelStatisticYCoCgSDR :191-222
            f3 m1 = splat3(0.0f), m2 = splat3(0.0f);
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                {
                    const f3 sdr = rgb_to_ycocg<YCOCG>(hdr_to_sdr(sample_curr(clampi(x + dx, 0, W - 1), clampi(y + dy, 0, H - 1))));
                    const float w = GAUSS ? std::exp(-3.0f * float(dx * dx + dy * dy) / ((1.0f + 1.0f) * (1.0f + 1.0f))) : 1.0f;
                    m1 += sdr * w;
                    m2 += sdr * sdr * w;
                    wsum += w;
                }
            const f3 mean = m1 / wsum;
            const f3 sd = sqrt3(max3(m2 / wsum - (mean * mean), 0.0f));
            // ClipToAABB :98-106
            const float maxT = 10.0f;
            const f3 ext = gamma * sd, dir = currY - prevY;
            const f3 sg{sign(dir.x), sign(dir.y), sign(dir.z)};
            const f3 isect = ((mean - sg * ext) - prevY) / dir;
            auto sel = [&](float i) { float ge = i >= 0.0f ? 1.0f : 0.0f; return (maxT + 1.0f) + ge * (i - (maxT + 1.0f)); };
            const float T = fmin2(maxT, fmin2(sel(isect.x), fmin2(sel(isect.y), sel(isect.z))));
            const float lt = T < maxT ? 1.0f : 0.0f;
            const f3 clamped = prevY + lt * ((prevY + dir * T) - prevY);
            const float alpha = prevRGBA.w * motionFactor * depthFactor;
            out.st4(x, y, mk4(sdr_to_hdr(ycocg_to_rgb<YCOCG>(lerp(currY, clamped, alpha))), corrected(alpha)));
        }
    return 0;
}

// ================================================================================================ Bloom
struct BloomAttribs { float Intensity, Threshold, SoftTreshold, Radius, AlphaInterpolation, p0, p1, p2; }; // BloomStructures.fxh:12-34
struct Taps13 { f3 A, B, C, D, E, F, G, H, I, J, K, L, M; };
inline f2 pixel_uv(int x, int y, int w, int h) { return ndc_to_uv({2.0f * ((float(x) + 0.5f) / float(w)) - 1.0f, 1.0f - 2.0f * ((float(y) + 0.5f) / float(h))}); }
inline Taps13 fetch13(const Img& in, f2 uv)
{
    const f2 ts{1.0f / float(in.w()), 1.0f / float(in.h())};
    auto S = [&](float ox, float oy) { return sample_linear_border3(in, uv.x + ts.x * ox, uv.y + ts.y * oy); };
    Taps13 t;
    t.A = S(-2, +2); t.B = S(0, +2); t.C = S(+2, +2); t.D = S(-2, 0); t.E = S(0, 0); t.F = S(+2, 0); t.G = S(-2, -2); t.H = S(0, -2); t.I = S(+2, -2);
    t.J = S(-1, +1); t.K = S(+1, +1); t.L = S(-1, -1); t.M = S(+1, -1);
    return t;
}

// ================================================================================================ PBR shading library
struct Srf { float rough; f3 r0, r90, diffuse; };
inline Srf surface_reflectance_workflow_mr(f3 base, float roughG, float metalB) // GetSurfaceReflectance, MR branch (PBR_Shading.fxh:376-426)
{
    Srf s;
    const f3 f0 = splat3(0.04f);
    s.diffuse = base * (splat3(1.0f) - f0) * (1.0f - metalB);
    const f3 spec = lerp(f0, base, metalB);
    s.rough = clampf(roughG, 0.0f, 1.0f);
    s.r0 = spec;
    s.r90 = splat3(clampf(max_comp(spec) * 50.0f, 0.0f, 1.0f));
    return s;
}
inline float perceived_brightness(f3 c) { return std::sqrt(0.299f * c.x * c.x + 0.587f * c.y * c.y + 0.114f * c.z * c.z); } // GetPerceivedBrightness (:93-96)
inline float solve_metallic(f3 diffuse, f3 specular, float oneMinusSpecularStrength)                                       // SolveMetallic (:99-117)
{
    const float minR = 0.04f;
    const float sb = perceived_brightness(specular);
    if (sb < minR) return 0.0f;
    const float db = perceived_brightness(diffuse);
    const float a = minR, b = db * oneMinusSpecularStrength / (1.0f - minR) + sb - 2.0f * minR, c = minR - sb;
    const float D = b * b - 4.0f * a * c;
    return clampf((-b + std::sqrt(D)) / (2.0f * a), 0.0f, 1.0f);
}
// GetSurfaceReflectance, specular-glossiness branch (:390-403) after ReadBaseLayerProperties (RenderPBR.psh:151-165): FastSRGBToLinear of the specular colour, factors = 1
inline Srf surface_reflectance_workflow_sg(f3 base, f4 desc, float& metallic)
{
    Srf s;
    const f3 f0{std::pow(desc.x, 2.2f), std::pow(desc.y, 2.2f), std::pow(desc.z, 2.2f)};
    s.rough = 1.0f - desc.w;
    const float oneMinus = 1.0f - std::max(std::max(f0.x, f0.y), f0.z);
    s.diffuse = base * oneMinus;
    metallic = solve_metallic(base, f0, oneMinus);
    s.rough = clampf(s.rough, 0.0f, 1.0f);
    const float r90 = clampf(std::max(std::max(f0.x, f0.y), f0.z) * 50.0f, 0.0f, 1.0f);
    s.r0 = f0;
    s.r90 = splat3(r90);
    return s;
}
inline Srf surface_reflectance_mr(f3 base, float metallic, float roughness) // GetSurfaceReflectanceMR (:429-449)
{
    Srf s;
    const float f0 = 0.04f;
    s.rough = roughness;
    s.diffuse = base * ((1.0f - f0) * (1.0f - metallic));
    s.r0 = lerp(splat3(f0), base, metallic);
    s.r90 = splat3(fmin2(max_comp(s.r0) * 50.0f, 1.0f));
    return s;
}
inline void smith_ggx_brdf(f3 toLight, f3 normal, f3 view, const Srf& srf, f3& diff, f3& spec, float& NdotL) // PBR_Common.fxh:340-405
{
    const f3 n = normalize(normal), v = normalize(view), l = normalize(toLight), h = normalize(l + v);
    NdotL = dot_sat(n, l);
    const float NdotV = dot_sat(n, v), NdotH = dot_sat(n, h), VdotH = dot_sat(v, h);
    diff = spec = splat3(0.0f);
    if (NdotL > 0.0f || NdotV > 0.0f)
    {
        const float alpha = srf.rough * srf.rough;
        const f3 F = schlick_reflection(VdotH, srf.r0, srf.r90);
        diff = (splat3(1.0f) - F) * (srf.diffuse / kPI);
        spec = F * smith_ggx_visibility_correlated(NdotL, NdotV, alpha) * normal_distribution_ggx(NdotH, alpha);
    }
}
struct Light // PBRLightAttribs, PBR_Structures.fxh:309-330
{
    int32_t Type; float PosX, PosY, PosZ, DirX, DirY, DirZ; int32_t ShadowMapIndex; float IntR, IntG, IntB, Range4, SpotScale, SpotOffset, p0, p1;
};
struct ShadeAttribs { float IBLScale[4]; float OcclusionStrength, EmissionScale, LastMip; int32_t LightCount; Light Lights[16]; int32_t Workflow, Padding[3]; };
// ---- shadow map of the punctual lights (ENABLE_SHADOWS): PBRShadowMapInfo (PBR_Structures.fxh:336-347), Texture2DArray<float> + Sam_ComparisonLinearClamp
struct ShadowInfo { float worldToLight[16]; float uvScale[2], uvBias[2]; float slice, pad0, pad1, pad2; };
static_assert(sizeof(ShadowInfo) == 96, "PBRShadowMapInfo layout");
struct Shadows
{
    const ref_args* a = nullptr; // in[9]: slices, in[10]: infos
    int pcf = 0;                 // PCF_FILTER_SIZE (2, 3, 5, 7); 0: shadows off
};
// SampleCmpLevelZero(Sam_ComparisonLinearClamp): bilinear blend of "reference < texel" over the clamped 2x2 footprint of the slice
inline float sample_cmp_level_zero(const Shadows& sh, float u, float v, float sliceF, float ref)
{
    const int n = sh.a->in_mips[9];
    const int s = clampi(int(std::floor(sliceF + 0.5f)), 0, n - 1);
    const Img im = in_img(sh.a, 9, s);
    const Bilinear b = bilinear_uc(u * float(im.w()), v * float(im.h()), im.w(), im.h());
    auto cmp = [&](int x, int y) { return ref < im.ld1(x, y) ? 1.0f : 0.0f; };
    return cmp(b.x0, b.y0) * b.w00 + cmp(b.x1, b.y0) * b.w10 + cmp(b.x0, b.y1) * b.w01 + cmp(b.x1, b.y1) * b.w11;
}
// FilterShadowMapFixedPCF (Shaders/Common/public/PCF.fxh:7-152), receiver-plane depth bias = 0 as ApplyPunctualLight passes it
inline float filter_shadow_map_fixed_pcf(const Shadows& sh, f2 uvIn, float slice, float depth)
{
    const Img im0 = in_img(sh.a, 9, 0);
    const f4 size{float(im0.w()), float(im0.h()), 1.0f / float(im0.w()), 1.0f / float(im0.h())};
    const f2 uv{uvIn.x * size.x, uvIn.y * size.y};
    f2 base{std::floor(uv.x + 0.5f), std::floor(uv.y + 0.5f)};
    const float s = uv.x + 0.5f - base.x, t = uv.y + 0.5f - base.y;
    base = f2{(base.x - 0.5f) * size.z, (base.y - 0.5f) * size.w};
    const float ref = fmax2(depth, 1e-8f); // DepthClamp
    auto S = [&](float u, float v) { return sample_cmp_level_zero(sh, base.x + u * size.z, base.y + v * size.w, slice, ref); };
    float sum = 0.0f;
    if (sh.pcf == 2) return sample_cmp_level_zero(sh, uvIn.x, uvIn.y, slice, ref);
    if (sh.pcf == 3)
    {
        const float uw0 = 3.0f - 2.0f * s, uw1 = 1.0f + 2.0f * s, u0 = (2.0f - s) / uw0 - 1.0f, u1 = s / uw1 + 1.0f;
        const float vw0 = 3.0f - 2.0f * t, vw1 = 1.0f + 2.0f * t, v0 = (2.0f - t) / vw0 - 1.0f, v1 = t / vw1 + 1.0f;
        sum += uw0 * vw0 * S(u0, v0); sum += uw1 * vw0 * S(u1, v0); sum += uw0 * vw1 * S(u0, v1); sum += uw1 * vw1 * S(u1, v1);
        return sum * 1.0f / 16.0f;
    }
    if (sh.pcf == 5)
    {
        const float uw[3] = {4.0f - 3.0f * s, 7.0f, 1.0f + 3.0f * s}, vw[3] = {4.0f - 3.0f * t, 7.0f, 1.0f + 3.0f * t};
        const float u[3] = {(3.0f - 2.0f * s) / uw[0] - 2.0f, (3.0f + s) / uw[1], s / uw[2] + 2.0f}, v[3] = {(3.0f - 2.0f * t) / vw[0] - 2.0f, (3.0f + t) / vw[1], t / vw[2] + 2.0f};
        for (int j = 0; j < 3; ++j)
            for (int i = 0; i < 3; ++i) sum += uw[i] * vw[j] * S(u[i], v[j]);
        return sum * 1.0f / 144.0f;
    }
    if (sh.pcf == 7)
    {
        const float uw[4] = {5.0f * s - 6.0f, 11.0f * s - 28.0f, -(11.0f * s + 17.0f), -(5.0f * s + 1.0f)}, vw[4] = {5.0f * t - 6.0f, 11.0f * t - 28.0f, -(11.0f * t + 17.0f), -(5.0f * t + 1.0f)};
        const float u[4] = {(4.0f * s - 5.0f) / uw[0] - 3.0f, (4.0f * s - 16.0f) / uw[1] - 1.0f, -(7.0f * s + 5.0f) / uw[2] + 1.0f, -s / uw[3] + 3.0f};
        const float v[4] = {(4.0f * t - 5.0f) / vw[0] - 3.0f, (4.0f * t - 16.0f) / vw[1] - 1.0f, -(7.0f * t + 5.0f) / vw[2] + 1.0f, -t / vw[3] + 3.0f};
        for (int j = 0; j < 4; ++j)
            for (int i = 0; i < 4; ++i) sum += uw[i] * vw[j] * S(u[i], v[j]);
        return sum * 1.0f / 2704.0f;
    }
    return 0.0f;
}
inline void apply_punctual_light(f3 pos, f3 normal, f3 view, const Srf& srf, const Light& L, f3& punctual, const Shadows& sh = Shadows{}) // ApplyPunctualLight (PBR_Shading.fxh:601-721)
{
    f3 dir{L.DirX, L.DirY, L.DirZ};
    float att = 1.0f;
    if (L.Type != 1)
    {
        f3 tp = pos - f3{L.PosX, L.PosY, L.PosZ};
        const float d2 = dot(tp, tp);
        tp = tp / std::sqrt(d2);
        float ra = 1.0f / d2;
        if (L.Range4 > 0.0f) ra *= sat(1.0f - (d2 * d2) / L.Range4);
        if (L.Type == 2) dir = tp;
        float ang = 1.0f;
        if (L.Type == 3) ang = sat(dot(tp, dir) * L.SpotScale + L.SpotOffset);
        att = ra * ang;
    }
    if (sh.pcf > 0 && L.ShadowMapIndex >= 0) // :644-660
    {
        ShadowInfo info;
        std::memcpy(&info, sh.a->in[10][0].data + size_t(L.ShadowMapIndex) * 24, sizeof(info));
        f4 sp = mul({pos.x, pos.y, pos.z, 1.0f}, info.worldToLight);
        sp.x /= sp.w; sp.y /= sp.w;
        const f2 uv = ndc_to_uv({sp.x, sp.y}) * f2{info.uvScale[0], info.uvScale[1]} + f2{info.uvBias[0], info.uvBias[1]};
        att *= filter_shadow_map_fixed_pcf(sh, uv, info.slice, sp.z);
    }
    if (att <= 0.0f) return;
    f3 diff, spec;
    float NdotL;
    smith_ggx_brdf(-dir, normal, view, srf, diff, spec, NdotL);
    punctual += (diff + spec) * (f3{L.IntR, L.IntG, L.IntB} * att) * NdotL;
}

// ---- software sampling of the IBL inputs (contract: oracle/ref/hlsl_shim.h hl_cube_*)
inline void cube_face_uv(f3 d, int& face, float& u, float& v)
{
    const float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    float ma, sc, tc;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az) { ma = ay; if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; } else { face = 3; sc = d.x; tc = -d.z; } }
    else { ma = az; if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    u = 0.5f * (sc / ma + 1.0f);
    v = 0.5f * (tc / ma + 1.0f);
}
inline f3 cube_dir(int face, float u, float v)
{
    const float sc = 2.0f * u - 1.0f, tc = 2.0f * v - 1.0f;
    switch (face)
    {
        case 0: return {1.f, -tc, -sc};
        case 1: return {-1.f, -tc, sc};
        case 2: return {sc, 1.f, tc};
        case 3: return {sc, -1.f, -tc};
        case 4: return {sc, -tc, 1.f};
        default: return {-sc, -tc, -1.f};
    }
}
inline f4 cube_texel(const Img& im, int face, int x, int y)
{
    const int n = im.w();
    if (x < 0 || y < 0 || x >= n || y >= n)
    {
        const f3 d = cube_dir(face, (float(x) + 0.5f) / float(n), (float(y) + 0.5f) / float(n));
        float u, v;
        cube_face_uv(d, face, u, v);
        x = clampi(int(std::floor(u * float(n))), 0, n - 1);
        y = clampi(int(std::floor(v * float(n))), 0, n - 1);
    }
    return im.ld4(x, face * n + y);
}
inline f4 cube_sample_level(const Img& im, f3 dir)
{
    int face; float u, v;
    cube_face_uv(dir, face, u, v);
    const int n = im.w();
    const float fx = u * float(n) - 0.5f, fy = v * float(n) - 0.5f;
    const float x0f = std::floor(fx), y0f = std::floor(fy), wx = fx - x0f, wy = fy - y0f;
    const int x0 = int(x0f), y0 = int(y0f);
    f4 acc = cube_texel(im, face, x0, y0) * ((1.0f - wx) * (1.0f - wy));
    acc += cube_texel(im, face, x0 + 1, y0) * (wx * (1.0f - wy));
    acc += cube_texel(im, face, x0, y0 + 1) * ((1.0f - wx) * wy);
    acc += cube_texel(im, face, x0 + 1, y0 + 1) * (wx * wy);
    return acc;
}
inline f4 cube_sample(const ref_args* a, int slot, f3 dir, float lod) // trilinear
{
    const int mips = a->in_mips[slot];
    lod = fmin2(fmax2(lod, 0.0f), float(mips - 1));
    const int l0 = int(std::floor(lod)), l1 = l0 + 1 < mips ? l0 + 1 : l0;
    const float f = lod - float(l0);
    const f4 c0 = cube_sample_level(in_img(a, slot, l0), dir);
    if (f == 0.0f || l1 == l0) return c0;
    const f4 c1 = cube_sample_level(in_img(a, slot, l1), dir);
    return c0 + (c1 - c0) * f;
}

// equirectangular ("sphere") environment map: Texture2D.SampleLevel(linear clamp, mip linear) at TransformDirectionToSphereMapUV(dir) (ShaderUtilities.fxh:98-102)
inline f2 direction_to_sphere_map_uv(f3 d)
{
    const float oneOverPi = 0.3183098862f;
    return {oneOverPi * (0.5f * std::atan2(d.z, d.x)) + 0.5f, oneOverPi * std::asin(d.y) + 0.5f};
}
inline f4 sphere_sample(const ref_args* a, int slot, f3 dir, float lod)
{
    const f2 uv = direction_to_sphere_map_uv(dir);
    const int mips = a->in_mips[slot];
    lod = clampf(lod, 0.0f, float(mips - 1));
    const int l0 = int(std::floor(lod)), l1 = l0 + 1 < mips ? l0 + 1 : l0;
    const float f = lod - float(l0);
    const f4 c0 = sample_linear_clamp4(in_img(a, slot, l0), uv.x, uv.y);
    if (f == 0.0f || l1 == l0) return c0;
    const f4 c1 = sample_linear_clamp4(in_img(a, slot, l1), uv.x, uv.y);
    return c0 + (c1 - c0) * f;
}
// environment lookup + solid angle of one texel for either map type (PBR_PrecomputeCommon.fxh:38-48); gamma: 1 in PrefilterEnvMap.psh:83, 0.5 in ComputeIrradianceMap.psh:71
inline f4 env_sample(const ref_args* a, bool sphere, f3 dir, float lod) { return sphere ? sphere_sample(a, 0, dir, lod) : cube_sample(a, 0, dir, lod); }
inline float env_pixel_solid_angle(bool sphere, float w, float h, f3 L, float gamma)
{
    if (!sphere) return 4.0f * kPI / (6.0f * w * h);
    const float theta = std::acos(L.y), dTheta = kPI / w, dPhi = 2.0f * kPI / h;
    return dPhi * (std::cos(theta - 0.5f * dTheta * gamma) - std::cos(theta + 0.5f * dTheta * gamma));
}
struct IBLInfo { f3 N, V, L; float NdotV; f2 preInt; f3 kS; };
inline IBLInfo ibl_sampling_info(const Srf& srf, const Img& lut, f3 N, f3 V) // GetIBLSamplingInfo (PBR_Shading.fxh:232-268)
{
    IBLInfo i;
    i.N = N; i.V = V;
    i.L = normalize(reflect(-V, N));
    i.NdotV = dot_sat(N, V);
    i.preInt = sample_linear_clamp2(lut, i.NdotV, srf.rough);
    i.kS = schlick_reflection(i.NdotV, srf.r0, max3(splat3(1.0f - srf.rough), srf.r0));
    return i;
}
inline f3 specular_ibl_ggx(const IBLInfo& i, f3 light) { return light * (i.kS * i.preInt.x + i.preInt.y); } // :293-304
inline f3 lambertian_ibl(const Srf& srf, const IBLInfo& i, f3 irr)                                           // :317-345
{
    const f3 FssEss = i.kS * i.preInt.x + i.preInt.y;
    const float Ems = 1.0f - (i.preInt.x + i.preInt.y);
    const f3 Favg = srf.r0 + (splat3(1.0f) - srf.r0) / 21.0f;
    const f3 Fms = FssEss * Favg / (splat3(1.0f) - Ems * Favg);
    const f3 kD = srf.diffuse * (splat3(1.0f) - (FssEss + Fms * Ems));
    return (Fms * Ems + kD) * irr;
}

// ---- IBL precompute helpers (PBR_PrecomputeCommon.fxh:10-42)
inline uint32_t reversebits(uint32_t v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
inline f2 hammersley2d(uint32_t i, uint32_t n) { return {float(i) / float(n), float(reversebits(i)) * 2.3283064365386963e-10f}; }
inline f3 importance_sample_ggx(f2 xi, float rough, f3 N)
{
    const float alpha = rough * rough, a2 = alpha * alpha;
    const float phi = 2.0f * kPI * xi.x;
    const float cosT = std::sqrt(sat((1.0f - xi.y) / (1.0f + (a2 - 1.0f) * xi.y)));
    const float sinT = std::sqrt(sat(1.0f - cosT * cosT));
    const f3 H{sinT * std::cos(phi), sinT * std::sin(phi), cosT};
    const f3 up = std::fabs(N.z) < 0.999f ? f3{0.f, 0.f, 1.f} : f3{1.f, 0.f, 0.f};
    const f3 tx = normalize(cross(up, N)), ty = cross(N, tx);
    return tx * H.x + ty * H.y + N * H.z;
}
inline float smith_ggx_sample_direction_pdf(f3 V, f3 N, f3 L, float alpha) // PBR_Common.fxh:297-324
{
    const f3 H = normalize(V + L);
    const float NdotH = dot(H, N), NdotV = dot(N, V), NdotL = dot(N, L);
    if (NdotH > 0.0f && NdotV > 0.0f && NdotL > 0.0f) return (smith_ggx_masking(NdotV, alpha) * normal_distribution_ggx(NdotH, alpha) / NdotV) / 4.0f;
    return 0.0f;
}
} // namespace

extern "C" {

// ------------------------------------------------------------------------------------------------ T1 (all eight feature-flag permutations, as in oracle/_ref:
// bit 0 = GAUSSIAN_WEIGHTING, bit 1 = BICUBIC_FILTER, bit 2 = YCOCG_COLOR_SPACE, TemporalAntiAliasing.hpp:62-74)
int oracle_taa_flags0(const ref_args* a) { return taa<false, false, false>(a); }
int oracle_taa_flags1(const ref_args* a) { return taa<true, false, false>(a); }
int oracle_taa_flags2(const ref_args* a) { return taa<false, true, false>(a); }
int oracle_taa_flags3(const ref_args* a) { return taa<true, true, false>(a); }
int oracle_taa_flags4(const ref_args* a) { return taa<false, false, true>(a); }
int oracle_taa_flags5(const ref_args* a) { return taa<true, false, true>(a); }
int oracle_taa_flags6(const ref_args* a) { return taa<false, true, true>(a); }
int oracle_taa_flags7(const ref_args* a) { return taa<true, true, true>(a); }

// ------------------------------------------------------------------------------------------------ B1: Bloom_ComputePrefilteredTexture.fx:19-85
int oracle_bloom_prefilter(const ref_args* a)
{
    BloomAttribs k;
    std::memcpy(&k, a->attribs, sizeof(k));
    const Img in = in_img(a, 0), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const Taps13 t = fetch13(in, pixel_uv(x, y, out.w(), out.h()));
            const float wts[5] = {0.125f, 0.125f, 0.125f, 0.125f, 0.5f};
            const f3 groups[5] = {(t.A + t.B + t.D + t.E) / 4.0f, (t.B + t.C + t.E + t.F) / 4.0f, (t.D + t.E + t.G + t.H) / 4.0f, (t.E + t.F + t.H + t.I) / 4.0f,
                                  (t.J + t.K + t.L + t.M) / 4.0f};
            f4 sum = splat4(0.0f);
            for (int g = 0; g < 5; ++g)
            {
                const float w = wts[g] * (1.0f / (1.0f + luminance601(groups[g])));
                sum += mk4(groups[g], 1.0f) * w;
            }
            const f3 color = xyz(sum) / (sum.w + 1.0e-5f);
            const float brightness = max_comp(color);
            const float knee = k.Threshold * k.SoftTreshold;
            float soft = clampf(brightness - k.Threshold + knee, 0.0f, 2.0f * knee);
            soft = soft * soft * 0.25f / (knee + 1.0e-5f);
            float contribution = fmax2(soft, brightness - k.Threshold);
            contribution /= fmax2(brightness, 1.0e-5f);
            out.st4(x, y, mk4(color * contribution, 0.0f));
        }
    return 0;
}
// B2: Bloom_ComputeDownsampledTexture.fx:11-44
int oracle_bloom_downsample(const ref_args* a)
{
    const Img in = in_img(a, 0), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const Taps13 t = fetch13(in, pixel_uv(x, y, out.w(), out.h()));
            f3 c = splat3(0.0f);
            c += (t.A + t.C + t.G + t.I) * 0.03125f;
            c += (t.B + t.D + t.F + t.H) * 0.0625f;
            c += (t.E + t.J + t.K + t.L + t.M) * 0.125f;
            out.st4(x, y, mk4(c, 0.0f));
        }
    return 0;
}
// B3: Bloom_ComputeUpsampledTexture.fx:20-55. in[0]: g_TextureInput, in[1]: g_TextureDownsampled; ival[0]: uInstID (!= 0: final composite)
int oracle_bloom_upsample(const ref_args* a)
{
    BloomAttribs k;
    std::memcpy(&k, a->attribs, sizeof(k));
    const Img input = in_img(a, 0), down = in_img(a, 1), out = out_img(a, 0);
    const bool final_pass = a->ival[0] != 0;
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const f2 uv = pixel_uv(x, y, out.w(), out.h());
            const f2 ts{1.0f / float(down.w()), 1.0f / float(down.h())};
            auto S = [&](float ox, float oy) { return xyz(sample_linear_clamp4(down, uv.x + ts.x * ox, uv.y + ts.y * oy)); };
            const f3 A = S(-1, +1), B = S(0, +1), C = S(+1, +1), D = S(-1, 0), E = S(0, 0), F = S(+1, 0), G = S(-1, -1), H = S(0, -1), I = S(+1, -1);
            f3 sum = E * 0.25f;
            sum += (B + D + F + H) * 0.125f;
            sum += (A + C + G + I) * 0.0625f;
            const f3 src = xyz(sample_linear_clamp4(input, uv.x, uv.y));
            if (final_pass) out.st4(x, y, mk4(lerp(src, src + k.Intensity * sum, k.AlphaInterpolation), input.ld4(x, y).w));
            else out.st4(x, y, mk4(src + sum, 0.0f));
        }
    return 0;
}

// ------------------------------------------------------------------------------------------------ P*: RenderPBR.psh lighting half from a G-buffer
// in: 0 base colour, 1 normal, 2 material (roughness, metallic), 3 depth, 4 emissive|none, 5 occlusion|none, 6 BRDF LUT, 7 irradiance cube, 8 prefiltered cube (mips)
int oracle_pbr_shade(const ref_args* a)
{
    set_depth_convention(a);
    const Camera cam = load_camera(a->cam0);
    ShadeAttribs sa;
    std::memcpy(&sa, a->attribs, sizeof(sa));
    const Img bc = in_img(a, 0), nrm = in_img(a, 1), mat = in_img(a, 2), depthTex = in_img(a, 3), lut = in_img(a, 6), o0 = out_img(a, 0), o1 = out_img(a, 1);
    const bool hasE = a->in_mips[4] > 0, hasAO = a->in_mips[5] > 0;
    const f3 camPos{cam.pos[0], cam.pos[1], cam.pos[2]};
    const Shadows shadows{a, a->ival[0]}; // ival[0]: PCF_FILTER_SIZE with ENABLE_SHADOWS (in[9] shadow-map slices, in[10] PBRShadowMapInfo array); 0 = no shadows
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < o0.h(); ++y)
        for (int x = 0; x < o0.w(); ++x)
        {
            const float depth = depthTex.ld1(x, y);
            if (is_background(depth))
            {
                o0.st4(x, y, {a->fval[0], a->fval[1], a->fval[2], a->fval[3]});
                if (o1.im->data) o1.st4(x, y, splat4(0.0f));
                continue;
            }
            const f4 base = bc.ld4(x, y), m = mat.ld4(x, y);
            const f3 N = nrm.ld3(x, y);
            const f3 pos = inv_project_position({(float(x) + 0.5f) * cam.viewport[2], (float(y) + 0.5f) * cam.viewport[3], depth}, cam.viewProjInv); // P9
            const f3 view = normalize(camPos - pos);                                                                                                   // RenderPBR.psh:309
            float metallicOut = 0.0f;
            const Srf srf = sa.Workflow == 1 ? surface_reflectance_workflow_sg(xyz(base), m, metallicOut)
                                             : surface_reflectance_workflow_mr(xyz(base), sat(m.x * 1.0f), sat(m.y * 1.0f)); // :138-184
            float occl = hasAO ? in_img(a, 5).ld1(x, y) : 1.0f;
            f3 emis = hasE ? in_img(a, 4).ld3(x, y) : splat3(0.0f);
            occl = lerp(1.0f, occl, sa.OcclusionStrength); // :311-316
            emis = emis * sa.EmissionScale;
            const f3 iblScale{sa.IBLScale[0], sa.IBLScale[1], sa.IBLScale[2]};
            f3 punctual = splat3(0.0f);
            const int nl = std::min(sa.LightCount, 16);
            for (int i = 0; i < nl; ++i) apply_punctual_light(pos, N, view, srf, sa.Lights[i], punctual, shadows);
            const IBLInfo ibl = ibl_sampling_info(srf, lut, N, view); // ApplyIBL (PBR_Shading.fxh:724-792)
            const f3 diffuseIBL = lambertian_ibl(srf, ibl, xyz(cube_sample(a, 7, ibl.N, 0.0f)));
            const f3 specularIBL = specular_ibl_ggx(ibl, xyz(cube_sample(a, 8, ibl.L, srf.rough * sa.LastMip)));
            const f3 color = punctual + (diffuseIBL + specularIBL) * iblScale * occl + emis; // ResolveLighting (:847-876)
            o0.st4(x, y, mk4(color, base.w));
            if (o1.im->data) o1.st4(x, y, mk4(specularIBL * iblScale * occl, 1.0f));
        }
    return 0;
}

// The Material target of a specular-glossiness surface (USD_Renderer.cpp:98).  in: 0 base colour, 1 PhysicalDesc (rgb specular sRGB, a glossiness); out: 0 (roughness, metallic, 0, 0)
int oracle_specgloss_material(const ref_args* a)
{
    const Img bc = in_img(a, 0), pd = in_img(a, 1), out = out_img(a, 0);
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            float metallic = 0.0f;
            const Srf s = surface_reflectance_workflow_sg(bc.ld3(x, y), pd.ld4(x, y), metallic);
            out.st4(x, y, {s.rough, metallic, 0.0f, 0.0f});
        }
    return 0;
}

// ------------------------------------------------------------------------------------------------ M1: HnPostProcess.psh:145-185
// in: 0 colour, 1 specular IBL, 2 SSR, 3 SSAO, 4 normal, 5 base colour, 6 material, 7 BRDF LUT; cam0; fval[0] SSRScale, fval[1] SSAOScale
int oracle_composite(const ref_args* a)
{
    const Camera cam = load_camera(a->cam0);
    const Img color = in_img(a, 0), sibl = in_img(a, 1), ssr = in_img(a, 2), ssao = in_img(a, 3), nrm = in_img(a, 4), bc = in_img(a, 5), mat = in_img(a, 6), lut = in_img(a, 7);
    const Img out = out_img(a, 0);
    const f3 camPos{cam.pos[0], cam.pos[1], cam.pos[2]};
#pragma omp parallel for
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const f4 c = color.ld4(x, y);
            f3 rgb = xyz(c);
            const float ssrScale = a->fval[0] * c.w;
            if (ssrScale > 0.0f)
            {
                const f4 refl = ssr.ld4(x, y), m = mat.ld4(x, y);
                const Srf srf = surface_reflectance_mr(bc.ld3(x, y), sat(m.y), sat(m.x));
                const f2 ndc{2.0f * (float(x) + 0.5f) / float(out.w()) - 1.0f, 1.0f - 2.0f * (float(y) + 0.5f) / float(out.h())};
                const f4 wp = mul({ndc.x, ndc.y, 0.5f, 1.0f}, cam.viewProjInv);
                const f3 view = normalize(camPos - xyz(wp) / wp.w);
                const IBLInfo ibl = ibl_sampling_info(srf, lut, nrm.ld3(x, y), view);
                rgb = rgb + (specular_ibl_ggx(ibl, xyz(refl)) - sibl.ld3(x, y)) * refl.w * ssrScale;
            }
            const float ssaoScale = a->fval[1] * c.w;
            if (ssaoScale > 0.0f) rgb = rgb * lerp(1.0f, ssao.ld1(x, y), ssaoScale);
            out.st4(x, y, mk4(rgb, c.w));
        }
    return 0;
}

// ------------------------------------------------------------------------------------------------ I1: PrecomputeBRDF.psh:8-50. out[0]: LUT (c=2); ival[0]: samples
int oracle_ibl_brdf_lut(const ref_args* a)
{
    const Img out = out_img(a, 0);
    const uint32_t n = uint32_t(a->ival[0]);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < out.h(); ++y)
        for (int x = 0; x < out.w(); ++x)
        {
            const float NoV = (float(x) + 0.5f) / float(out.w()), rough = (float(y) + 0.5f) / float(out.h());
            const f3 V{std::sqrt(1.0f - NoV * NoV), 0.0f, NoV}, N{0.f, 0.f, 1.f};
            float A = 0.0f, B = 0.0f;
            for (uint32_t i = 0; i < n; ++i)
            {
                const f3 H = importance_sample_ggx(hammersley2d(i, n), rough, N);
                const f3 L = 2.0f * dot(V, H) * H - V;
                const float NoL = sat(L.z), NoH = sat(H.z), VoH = sat(dot(V, H));
                if (NoL > 0.0f)
                {
                    const float gvis = 4.0f * smith_ggx_visibility_correlated(NoL, NoV, rough * rough) * VoH * NoL / NoH;
                    const float fc = std::pow(1.0f - VoH, 5.0f);
                    A += (1.0f - fc) * gvis;
                    B += fc * gvis;
                }
            }
            out.st2(x, y, {A / float(n), B / float(n)});
        }
    return 0;
}
// I2: PrefilterEnvMap.psh:40-98. in[0]: environment cube (mips); out[0]: one mip (w x 6w); fval[0]: roughness; ival[0]: samples
int oracle_ibl_prefilter_env_map(const ref_args* a)
{
    const Img out = out_img(a, 0);
    const int n = out.w();
    const bool sphere = a->ival[1] != 0; // ENV_MAP_TYPE_SPHERE: in[0] is a 2D mip chain
    const float roughness = a->fval[0], envW = float(a->in[0][0].w), envH = sphere ? float(a->in[0][0].h) : envW, mipCount = float(a->in_mips[0]);
    const uint32_t ns = uint32_t(a->ival[0]);
#pragma omp parallel for schedule(dynamic, 1)
    for (int row = 0; row < 6 * n; ++row)
        for (int x = 0; x < n; ++x)
        {
            const f3 R = normalize(cube_dir(row / n, (float(x) + 0.5f) / float(n), (float(row % n) + 0.5f) / float(n)));
            const f3 N = R, V = R;
            f3 color = splat3(0.0f);
            float total = 0.0f;
            for (uint32_t i = 0; i < ns; ++i)
            {
                const f3 H = importance_sample_ggx(hammersley2d(i, ns), roughness, N);
                const f3 L = 2.0f * dot(V, H) * H - V;
                const float NoL = clampf(dot(N, L), 0.0f, 1.0f), VoH = clampf(dot(V, H), 0.0f, 1.0f);
                if (NoL > 0.0f && VoH > 0.0f)
                {
                    const float alpha = roughness * roughness;
                    const float pdf = fmax2(smith_ggx_sample_direction_pdf(V, N, L, alpha), 0.0001f);
                    const float omegaS = 1.0f / (float(ns) * pdf), omegaP = env_pixel_solid_angle(sphere, envW, envH, L, 1.0f);
                    const float mip = (alpha == 0.0f) ? 0.0f : clampf(0.5f * std::log2(omegaS / fmax2(omegaP, 1e-10f)) + 1.0f, 0.0f, mipCount - 1.0f);
                    color += xyz(env_sample(a, sphere, L, mip)) * NoL;
                    total += NoL;
                }
            }
            out.st4(x, row, mk4(color / total, 0.0f));
        }
    return 0;
}
// I3: ComputeIrradianceMap.psh:43-83. in[0]: environment cube (mips); out[0]: irradiance cube; ival[0]: samples
int oracle_ibl_irradiance_map(const ref_args* a)
{
    const Img out = out_img(a, 0);
    const int n = out.w();
    const bool sphere = a->ival[1] != 0; // ENV_MAP_TYPE_SPHERE
    const float envW = float(a->in[0][0].w), envH = sphere ? float(a->in[0][0].h) : envW, mipCount = float(a->in_mips[0]);
    const uint32_t ns = uint32_t(a->ival[0]);
#pragma omp parallel for schedule(dynamic, 1)
    for (int row = 0; row < 6 * n; ++row)
        for (int x = 0; x < n; ++x)
        {
            const f3 N = normalize(cube_dir(row / n, (float(x) + 0.5f) / float(n), (float(row % n) + 0.5f) / float(n)));
            const f3 T = normalize(cross(N, std::fabs(N.y) > 0.5f ? f3{1.f, 0.f, 0.f} : f3{0.f, 1.f, 0.f})), B = cross(T, N);
            f3 irr = splat3(0.0f);
            for (uint32_t i = 0; i < ns; ++i)
            {
                const f2 xi = hammersley2d(i, ns);
                f3 L{std::cos(2.0f * kPI * xi.x) * std::sqrt(1.0f - xi.y), std::sin(2.0f * kPI * xi.x) * std::sqrt(1.0f - xi.y), std::sqrt(xi.y)};
                const float pdf = fmax2(L.z, 1e-6f) / kPI;
                L = normalize(L.x * T + L.y * B + L.z * N);
                const float omegaS = 1.0f / (float(ns) * pdf), omegaP = env_pixel_solid_angle(sphere, envW, envH, L, 0.5f);
                irr += xyz(env_sample(a, sphere, L, clampf(0.5f * std::log2(omegaS / fmax2(omegaP, 1e-10f)) + 1.0f, 0.0f, mipCount - 1.0f)));
            }
            out.st4(x, row, mk4(irr / float(ns), 1.0f));
        }
    return 0;
}

} // extern "C"

// ------------------------------------------------------------------------------------------------ auto exposure
// TEST INFRASTRUCTURE.  Low-resolution luminance (UnwarpEpipolarScattering.fx:283-307 without in-scattering / extinction; GetWeightedLogLum,
// AtmosphereShadersCommon.fxh:197-203), GenerateMips (2x2 box per level) of the 64x64 image down to 1x1, UpdateAverageLuminancePS
// (UpdateAverageLuminance.fx:12-29) blended with BS_AlphaBlend (EpipolarLightScattering.cpp:1827).
// in[0]: scene colour (c=4); out[0]: low-res luminance 64x64 (c=2); out[1]: average luminance 1x1 (c=1, read-modify-write);
// fval[0]: elapsed time; ival[0]: LIGHT_ADAPTATION
extern "C" int oracle_autoexposure(const ref_args* a)
{
    const Img color{&a->in[0][0]};
    const ref_img& low = a->out[0];
    if (low.w != 64 || low.h != 64 || low.c != 2) return -1;
    std::vector<f2> lvl(64 * 64);
    for (int y = 0; y < 64; ++y)
        for (int x = 0; x < 64; ++x)
        {
            const f4    c   = sample_linear_clamp4(color, (float(x) + 0.5f) / 64.0f, (float(y) + 0.5f) / 64.0f);
            const float lum = dot(xyz(c), f3{0.212671f, 0.715160f, 0.072169f});
            const float w   = sat((lum - 0.01f) / 0.01f);
            const f2    q   = {std::log(fmax2(lum, 1e-5f)) * w, w};
            lvl[size_t(y) * 64 + x] = q;
            float* o = low.data + (size_t(y) * 64 + x) * 2;
            o[0] = q.x; o[1] = q.y;
        }
    for (int n = 64; n > 1; n /= 2) // GenerateMips: every level is the 2x2 box filter of the previous one
    {
        std::vector<f2> nxt(size_t(n / 2) * (n / 2));
        for (int y = 0; y < n / 2; ++y)
            for (int x = 0; x < n / 2; ++x)
            {
                const f2 p00 = lvl[size_t(2 * y) * n + 2 * x], p10 = lvl[size_t(2 * y) * n + 2 * x + 1], p01 = lvl[size_t(2 * y + 1) * n + 2 * x],
                         p11 = lvl[size_t(2 * y + 1) * n + 2 * x + 1];
                nxt[size_t(y) * (n / 2) + x] = {((p00.x + p10.x) + (p01.x + p11.x)) * 0.25f, ((p00.y + p10.y) + (p01.y + p11.y)) * 0.25f};
            }
        lvl.swap(nxt);
    }
    float newWeight = a->ival[0] ? 1.0f - std::exp(-1.0f * a->fval[0]) : 1.0f;
    const float logLum = lvl[0].x / fmax2(lvl[0].y, 1e-6f);
    newWeight *= sat(lvl[0].y / 1e-3f);
    float* avg = a->out[1].data;
    *avg = std::exp(logLum) * newWeight + *avg * (1.0f - newWeight);
    return 0;
}

// ------------------------------------------------------------------------------------------------ E1: environment-map background
// TEST INFRASTRUCTURE.  Shaders/Common/private/EnvMap.psh:46-77 (SampleEnvMap) behind EnvMap.vsh:9-23 and the pipeline state of
// Components/src/EnvMapRenderer.cpp:176-183 (linear-clamp sampler, depth test LESS_EQUAL at the far plane, no depth writes).
// in: 0 environment cube (mips), 1 depth; cam0 / cam1; attribs = ToneMappingAttribs; fval: 0 AverageLogLum, 1 MipLevel, 2 Alpha, 3..5 Scale;
// ival: 0 CONVERT_OUTPUT_TO_SRGB, 1 COMPUTE_MOTION_VECTORS, 2 ENV_MAP_TYPE_SPHERE, 7 USE_REVERSE_DEPTH.  out: 0 colour, 1 motion -- written where the depth test passes.
extern "C" int oracle_tonemap(const ref_args* a);
extern "C" int oracle_envmap(const ref_args* a)
{
    const Camera cam = load_camera(a->cam0), prev = load_camera(a->cam1);
    const Img depth = in_img(a, 1), o0 = out_img(a, 0), o1 = out_img(a, 1);
    const float mip = a->fval[1], alpha = a->fval[2];
    const f3 scale{a->fval[3], a->fval[4], a->fval[5]};
    const bool gamma = a->ival[0] != 0, motionVectors = a->ival[1] != 0;
    const bool sphere = a->ival[2] != 0;   // ENV_MAP_TYPE_SPHERE: in[0] is a 2D mip chain
    const bool reversed = a->ival[7] != 0; // OPTION_FLAG_USE_REVERSE_DEPTH: COMPARISON_FUNC_GREATER_EQUAL (EnvMapRenderer.cpp:182)
    auto passes = [&](float d) { return reversed ? cam.farDepth >= d : cam.farDepth <= d; };
    int32_t mode = 0;
    std::memcpy(&mode, a->attribs, sizeof(mode)); // ToneMappingAttribs::iToneMappingMode
    const int W = o0.w(), H = o0.h();
    std::vector<float> hdr(size_t(W) * H * 4, 0.0f);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
        {
            if (!passes(depth.ld1(x, y))) continue;
            const float u = (float(x) + 0.5f) / float(W), v = (float(y) + 0.5f) / float(H);
            const f4 clip{2.0f * u - 1.0f, 1.0f - 2.0f * v, cam.farDepth, 1.0f};
            const f4 world = mul(clip, cam.viewProjInv);
            const f3 dir = f3{world.x, world.y, world.z} / world.w - f3{cam.pos[0], cam.pos[1], cam.pos[2]};
            const f3 c = xyz(env_sample(a, sphere, normalize(dir), mip)) * scale;
            float* h = &hdr[(size_t(y) * W + x) * 4];
            h[0] = c.x; h[1] = c.y; h[2] = c.z; h[3] = alpha;
            if (motionVectors) // :61-66
            {
                const f3 prevWorld = f3{prev.pos[0], prev.pos[1], prev.pos[2]} + dir;
                f4 prevClip = mul({prevWorld.x, prevWorld.y, prevWorld.z, 1.0f}, prev.viewProj);
                prevClip.x /= prevClip.w; prevClip.y /= prevClip.w;
                o1.st2(x, y, {(clip.x - cam.jitter[0]) - (prevClip.x - prev.jitter[0]), (clip.y - cam.jitter[1]) - (prevClip.y - prev.jitter[1])}); // GetMotionVector
            }
            else
                o1.st2(x, y, {0.0f, 0.0f});
        }
    if (mode > 0) // ToneMap(Color.rgb, g_ToneMappingAttribs, g_AverageLogLum) :53-55, through the oracle's tone-map entry on the whole buffer
    {
        std::vector<float> ldr(hdr.size());
        ref_args t{};
        t.in[0][0]   = ref_img{hdr.data(), W, H, 4};
        t.in_mips[0] = 1;
        t.out[0]     = ref_img{ldr.data(), W, H, 4};
        t.attribs    = a->attribs;
        t.fval[0]    = a->fval[0];
        t.ival[0]    = 0;
        if (oracle_tonemap(&t) != 0) return -1;
        hdr.swap(ldr);
    }
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
        {
            if (!passes(depth.ld1(x, y))) continue;
            const float* h = &hdr[(size_t(y) * W + x) * 4];
            f3 c{h[0], h[1], h[2]};
            if (gamma) c = {std::pow(c.x, 1.0f / 2.2f), std::pow(c.y, 1.0f / 2.2f), std::pow(c.z, 1.0f / 2.2f)};
            o0.st4(x, y, mk4(c, alpha));
        }
    return 0;
}
