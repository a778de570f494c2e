// mifx_pbr_layers.h -- the material extensions of the reference's PBR lighting library: clear coat, sheen, anisotropy, iridescence, transmission
// (ENABLE_CLEAR_COAT / ENABLE_SHEEN / ENABLE_ANISOTROPY / ENABLE_IRIDESCENCE / ENABLE_TRANSMISSION of Shaders/PBR/public/PBR_Shading.fxh:40-62, compiled out of
// the default permutation: PBR/interface/PBR_Renderer.hpp:159-179).  Follows Shaders/Common/public/PBR_Common.fxh:91-103,126-136,197-209,407-509,
// Shaders/PBR/private/Iridescence.fxh and the ENABLE_* blocks of PBR_Shading.fxh:122-142,232-291,347-368,452-467,601-876.
//
// Not the timed path of the chain (the kernel that includes this is launched only when a frame carries one of the layers; the default shade kernel is untouched): libm
// exp / pow / cos throughout, fdiv / fsqrt (mifx_device.h: within an ulp in 4 million of the correctly rounded results; the IEEE operations when compiled for the host).
// Operations are written in the reference's order: the checker is the reference's own source.
#pragma once
#include "mifx_pbr.h"

namespace mifx
{
MIFX_D float schlick_reflection1(float VdotH, float r0, float r90) { return r0 + (r90 - r0) * pow5(clampf(1.0f - VdotH, 0.0f, 1.0f)); } // PBR_Common.fxh:82-85
MIFX_D v3    schlick_to_f0(float VdotH, v3 f, v3 f90)                                                                                    // :98-103
{
    const float x  = clampf(1.0f - VdotH, 0.0f, 1.0f);
    const float x5 = clampf(pow5(x), 0.0f, 0.9999f);
    return (f - f90 * x5) / (1.0f - x5);
}

// ---- clear coat: GetSurfaceReflectanceClearCoat (PBR_Shading.fxh:452-467)
MIFX_D SurfaceReflectance surface_reflectance_clear_coat(float roughness, float ior)
{
    SurfaceReflectance s;
    float f0 = fdiv(ior - 1.0f, ior + 1.0f);
    f0 *= f0;
    s.perceptualRoughness = roughness;
    s.diffuse = mk3(0.0f);
    s.r0      = mk3(f0);
    s.r90     = mk3(1.0f);
    return s;
}

// ---- sheen ("Production Friendly Microfacet Sheen BRDF", Estevez and Kulla 2017): PBR_Common.fxh:458-509
MIFX_D float normal_distribution_charlie(float NdotH, float sheenRoughness)
{
    sheenRoughness    = fmaxf(sheenRoughness, 1e-6f);
    const float alpha = sheenRoughness * sheenRoughness;
    const float invA  = fdiv(1.0f, alpha);
    const float cos2h = NdotH * NdotH;
    const float sin2h = fmaxf(1.0f - cos2h, 0.0078125f);
    return fdiv((2.0f + invA) * powf(sin2h, invA * 0.5f), 2.0f * MIFX_PI);
}
// The light-independent part of SheenVisibility / LambdaSheen, evaluated once per pixel instead of once per light (the same values: the five fitted coefficients depend on the
// roughness only, Lambda(NdotV) on the view, and the numeric helper at 0.5 is a constant of the pixel) -- a pow and two exp per light less.
struct SheenFrame
{
    float a, b, c, d, e; // LambdaSheenNumericHelper's coefficients for this roughness (:470-479)
    float helperHalf;    // LambdaSheenNumericHelper(0.5, AlphaG)
    float lambdaV;       // LambdaSheen(NdotV, AlphaG)
    float NdotV;
};
MIFX_D float lambda_sheen_numeric_helper(const SheenFrame& f, float x) { return fdiv(f.a, 1.0f + f.b * powf(x, f.c)) + f.d * x + f.e; }
MIFX_D float lambda_sheen(const SheenFrame& f, float cosTheta) // :481-491
{
    if (fabsf(cosTheta) < 0.5f) return expf(lambda_sheen_numeric_helper(f, cosTheta));
    return expf(2.0f * f.helperHalf - lambda_sheen_numeric_helper(f, 1.0f - cosTheta));
}
MIFX_D SheenFrame sheen_frame(float sheenRoughness, float NdotV)
{
    sheenRoughness     = fmaxf(sheenRoughness, 1e-6f); // SheenVisibility (:493-502)
    const float alphaG = sheenRoughness * sheenRoughness;
    const float t = (1.0f - alphaG) * (1.0f - alphaG);
    SheenFrame f;
    f.a = lerpf(21.5473f, 25.32450f, t);
    f.b = lerpf(3.82987f, 3.32435f, t);
    f.c = lerpf(0.19823f, 0.16801f, t);
    f.d = lerpf(-1.97760f, -1.27393f, t);
    f.e = lerpf(-4.32054f, -4.85967f, t);
    f.helperHalf = lambda_sheen_numeric_helper(f, 0.5f);
    f.NdotV      = NdotV;
    f.lambdaV    = lambda_sheen(f, NdotV);
    return f;
}
MIFX_D float sheen_visibility(const SheenFrame& f, float NdotL)
{
    const float eps = 5e-8f;
    return saturate(fdiv(1.0f, (1.0f + f.lambdaV + lambda_sheen(f, NdotL)) * fmaxf(4.0f * f.NdotV * NdotL, eps)));
}
MIFX_D v3 sheen_specular_brdf(const SheenFrame& f, v3 sheenColor, float sheenRoughness, float NdotL, float NdotH) // :504-509
{
    const float D   = normal_distribution_charlie(NdotH, sheenRoughness);
    const float Vis = sheen_visibility(f, NdotL);
    return sheenColor * D * Vis;
}
// ApplyDirectionalLightSheen (PBR_Shading.fxh:133-142): the normal and the view vector as they are (not re-normalised); f = sheen_frame(roughness, dot_sat(N, V))
MIFX_D v3 apply_directional_light_sheen(const SheenFrame& f, v3 lightDir, v3 lightColor, v3 sheenColor, float sheenRoughness, v3 N, v3 V)
{
    const v3    L = -lightDir;
    const v3    H = normalize(V + L);
    const float NdotL = dot_sat(N, L), NdotH = dot_sat(N, H);
    return lightColor * NdotL * sheen_specular_brdf(f, sheenColor, sheenRoughness, NdotL, NdotH);
}
// one channel of a look-up table (.Sample(Sam_LinearClamp).r): the sheen albedo-scaling table and the preintegrated Charlie BRDF
MIFX_D float lut_sample_r(const LutK& t, float u, float v)
{
    const Bilinear b = bilinear_uc(u * float(t.size_w), v * float(t.size_h), t.size_w, t.size_h);
    auto ldp = [&](int x, int y) { return t.data[size_t(y) * t.pitch_f + size_t(x) * t.comps]; };
    return ldp(b.x0, b.y0) * b.w00 + ldp(b.x1, b.y0) * b.w10 + ldp(b.x0, b.y1) * b.w01 + ldp(b.x1, b.y1) * b.w11;
}

// ---- anisotropy: NormalDistribution_GGX_Anisotropic (PBR_Common.fxh:197-209), SmithGGXVisibilityCorrelated_Anisotropic (:126-136), SmithGGX_BRDF_Anisotropic (:407-455)
struct AnisotropyInfo // AnisotropyShadingInfo (PBR_Shading.fxh:506-514)
{
    float strength;
    v3    tangent, bitangent;
    float alphaT, alphaB;
};
MIFX_D void smith_ggx_brdf_anisotropic(v3 pointToLight, v3 normal, v3 view, const AnisotropyInfo& an, const SurfaceReflectance& srf, v3& diffuse, v3& spec, float& NdotL)
{
    const v3 n = normalize(normal), v = normalize(view), l = normalize(pointToLight), h = normalize(l + v); // GetAngularInfo (:340-360)
    NdotL = dot_sat(n, l);
    const float NdotV = dot_sat(n, v), NdotH = dot_sat(n, h), VdotH = dot_sat(v, h);
    diffuse = mk3(0.0f);
    spec    = mk3(0.0f);
    if (NdotL > 0.0f || NdotV > 0.0f)
    {
        const float TdotH = dot(an.tangent, h), BdotH = dot(an.bitangent, h), TdotL = dot(an.tangent, l), TdotV = dot(an.tangent, v), BdotL = dot(an.bitangent, l),
                    BdotV = dot(an.bitangent, v);
        const float a2 = an.alphaT * an.alphaB;
        const v3    dv{an.alphaB * TdotH, an.alphaT * BdotH, a2 * NdotH};
        const float w2 = fdiv(a2, fmaxf(dot(dv, dv), 1e-6f));
        const float D  = a2 * w2 * w2 * (1.0f / MIFX_PI);
        const float lambdaV = NdotL * fmaxf(length(v3{an.alphaT * TdotV, an.alphaB * BdotV, NdotV}), 1e-3f);
        const float lambdaL = NdotV * fmaxf(length(v3{an.alphaT * TdotL, an.alphaB * BdotL, NdotL}), 1e-3f);
        const float Vis = fdiv(0.5f, lambdaV + lambdaL);
        const v3    F   = schlick_reflection(VdotH, srf.r0, srf.r90);
        diffuse = (mk3(1.0f) - F) * (srf.diffuse / MIFX_PI);
        spec    = F * Vis * D;
    }
}

// ---- iridescence: Shaders/PBR/private/Iridescence.fxh (Belcour and Barla 2017, as in the glTF sample viewer)
MIFX_D float sqr(float v) { return v * v; }
MIFX_D v3    sqr(v3 v) { return v * v; }
MIFX_D v3    fresnel0_to_ior(v3 f0) // :6-10
{
    const v3 s = sqrt3(f0);
    return (mk3(1.0f) + s) / (mk3(1.0f) - s);
}
MIFX_D v3    ior_to_fresnel0(v3 transmittedIor, float incidentIor) { return sqr((transmittedIor - mk3(incidentIor)) / (transmittedIor + mk3(incidentIor))); } // :16-21
MIFX_D float ior_to_fresnel0(float transmittedIor, float incidentIor) { return sqr(fdiv(transmittedIor - incidentIor, transmittedIor + incidentIor)); }         // :24-27
MIFX_D v3    eval_sensitivity(float opd, v3 shift)                                                                                                             // :32-51
{
    const float phase = 2.0f * MIFX_PI * opd * 1.0e-9f;
    const v3 val{5.4856e-13f, 4.4201e-13f, 5.2481e-13f};
    const v3 pos{1.6810e+06f, 1.7953e+06f, 2.2084e+06f};
    const v3 var{4.3278e+09f, 9.3046e+09f, 6.6121e+09f};
    const v3 arg = pos * phase + shift;
    const v3 dmp = -sqr(phase) * var;
    v3 xyz = val * sqrt3(2.0f * MIFX_PI * var) * v3{cosf(arg.x), cosf(arg.y), cosf(arg.z)} * v3{expf(dmp.x), expf(dmp.y), expf(dmp.z)};
    xyz.x += 9.7470e-14f * fsqrt(2.0f * MIFX_PI * 4.5282e+09f) * cosf(2.2399e+06f * phase + shift.x) * expf(-4.5282e+09f * sqr(phase));
    xyz = xyz / 1.0685e-7f;
    return v3{3.2404542f * xyz.x - 1.5371385f * xyz.y - 0.4985314f * xyz.z, -0.9692660f * xyz.x + 1.8760108f * xyz.y + 0.0415560f * xyz.z,
              0.0556434f * xyz.x - 0.2040259f * xyz.y + 1.0572252f * xyz.z};
}
MIFX_D float smoothstep1(float a, float b, float x)
{
    const float t = saturate(fdiv(x - a, b - a));
    return t * t * (3.0f - 2.0f * t);
}
MIFX_D v3 eval_iridescence(float outsideIor, float eta2, float cosTheta1, float thickness, v3 baseF0) // :53-111
{
    const float iridescenceIor = lerpf(outsideIor, eta2, smoothstep1(0.0f, 0.03f, thickness));
    const float sinTheta2Sq = sqr(fdiv(outsideIor, iridescenceIor)) * (1.0f - sqr(cosTheta1));
    const float cosTheta2Sq = 1.0f - sinTheta2Sq;
    if (cosTheta2Sq < 0.0f) return mk3(1.0f); // total internal reflection
    const float cosTheta2 = fsqrt(cosTheta2Sq);
    // first interface
    const float R0   = ior_to_fresnel0(iridescenceIor, outsideIor);
    const float R12  = schlick_reflection1(cosTheta1, R0, 1.0f);
    const float T121 = 1.0f - R12;
    float phi12 = 0.0f;
    if (iridescenceIor < outsideIor) phi12 = MIFX_PI;
    const float phi21 = MIFX_PI - phi12;
    // second interface
    const v3 baseIor = fresnel0_to_ior(v3{clampf(baseF0.x, 0.0f, 0.9999f), clampf(baseF0.y, 0.0f, 0.9999f), clampf(baseF0.z, 0.0f, 0.9999f)});
    const v3 R1  = ior_to_fresnel0(baseIor, iridescenceIor);
    const v3 R23 = schlick_reflection(cosTheta2, R1, mk3(1.0f));
    const v3 phi23{baseIor.x < iridescenceIor ? MIFX_PI : 0.0f, baseIor.y < iridescenceIor ? MIFX_PI : 0.0f, baseIor.z < iridescenceIor ? MIFX_PI : 0.0f};
    // phase shift
    const float opd = 2.0f * iridescenceIor * thickness * cosTheta2;
    const v3    phi = mk3(phi21) + phi23;
    // compound terms
    const v3 R123r = R12 * R23;
    const v3 R123{clampf(R123r.x, 1e-5f, 0.9999f), clampf(R123r.y, 1e-5f, 0.9999f), clampf(R123r.z, 1e-5f, 0.9999f)};
    const v3 r123 = sqrt3(R123);
    const v3 Rs   = sqr(T121) * R23 / (mk3(1.0f) - R123);
    v3 I  = mk3(R12) + Rs; // m = 0
    v3 Cm = Rs - mk3(T121);
    for (int m = 1; m <= 2; ++m)
    {
        Cm = Cm * r123;
        const v3 Sm = 2.0f * eval_sensitivity(float(m) * opd, float(m) * phi);
        I = I + Cm * Sm;
    }
    return max3(I, mk3(0.0f));
}

// ---- the kernel's per-pixel body
struct LayersK
{
    unsigned flags; // MIFX_PBR_LAYER_*
    float    iridescenceIor, rotationCos, rotationSin;
    int      hasClearcoatNormal, hasTangent;
    Img      clearcoat, clearcoatNormal, sheen, anisotropy, tangent, iridescence, transmission;
    LutK     albedoScaling, charlie;
};
// One pixel of the shade with material layers (the body of pbr_shade_layers_kernel and of the sharded hit fetch pbr_hit_fetch_layers_kernel, pbr.hip): colour and
// specular IBL of pixel (x, y), nothing stored.  APRON: the cube maps are the apron copies of cube_apron_kernel (the kernel);
// false = plain face arrays (tests/host_kernels compiles this function for the host and runs it without the copies).
// SET: the layer set as a compile-time constant (the launcher instantiates each single layer and all five: the branches fold, and the permutation costs what it uses), or
// kLayersRuntime: the set is read from LayersK::flags (any other combination; uniform branches, but the register budget of all five layers).  SHADOWS = ENABLE_SHADOWS
// (the PCF filter alone costs ~55 registers: a template parameter as in the default shade).
constexpr unsigned kLayersRuntime = 0xffffffffu;
template <bool APRON, unsigned SET, bool SHADOWS>
MIFX_D void pbr_shade_layers_pixel(int x, int y, const Img& baseColor, const Img& normalTex, const Img& material, const Img& depthTex, const Img& emissive, const Img& occlusion,
                                   const LutK& lut, const v4* irradiance0, int irradianceSize, const v4* const* prefMips, int prefSize, int prefLevels, const CamK& cam,
                                   const ShadeK& k, const LayersK& ly, int hasEmissive, int hasAo, const ShadowK& sh, v4& outColor, v4& outSpec)
{
    auto irradianceAt = [&](v3 d) { return APRON ? cube_sample_level_apron(irradiance0, irradianceSize, d) : cube_sample_level(irradiance0, irradianceSize, d); };
    auto prefilteredAt = [&](v3 d, float lod) { return APRON ? cube_sample_apron(prefMips, prefSize, prefLevels, d, lod) : cube_sample(prefMips, prefSize, prefLevels, d, lod); };
    const float depth = ld<float>(depthTex, x, y);
    if (is_background(depth, cam.reversedDepth != 0))
    {
        outColor = v4{k.background[0], k.background[1], k.background[2], k.background[3]};
        outSpec  = mk4(0.0f);
        return;
    }
    const unsigned set = SET == kLayersRuntime ? ly.flags : SET;
    const bool clearCoat = (set & MIFX_PBR_LAYER_CLEAR_COAT) != 0u, sheen = (set & MIFX_PBR_LAYER_SHEEN) != 0u, aniso = (set & MIFX_PBR_LAYER_ANISOTROPY) != 0u;
    const bool irid = (set & MIFX_PBR_LAYER_IRIDESCENCE) != 0u, transm = (set & MIFX_PBR_LAYER_TRANSMISSION) != 0u;
    const v4 bc  = ld<v4>(baseColor, x, y);
    const v4 mat = ld<v4>(material, x, y);
    const v3 N   = xyz(ld<v4>(normalTex, x, y));
    const v3 pos  = inv_project_position(v3{(float(x) + 0.5f) * cam.ivw, (float(y) + 0.5f) * cam.ivh, depth}, cam.viewProjInv);
    const v3 view = normalize(v3{cam.pos[0], cam.pos[1], cam.pos[2]} - pos);
    float unusedMetallic;
    SurfaceReflectance srf = k.workflow == MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS ? surface_reflectance_workflow_sg(xyz(bc), mat, unusedMetallic)
                                                                                 : surface_reflectance_workflow_mr(xyz(bc), saturate(mat.x * 1.0f), saturate(mat.y * 1.0f));
    const float baseNdotV = dot_sat(N, view); // BaseLayer.NdotV (RenderPBR.psh:181)
    float occl = hasAo ? ld<float>(occlusion, x, y) : 1.0f;
    v3    emis = hasEmissive ? xyz(ld<v4>(emissive, x, y)) : mk3(0.0f);
    occl = lerpf(1.0f, occl, k.occlusionStrength);
    emis = emis * k.emissionScale;
    const v3 iblScale{k.iblScale[0], k.iblScale[1], k.iblScale[2]};

    // GetSurfaceShadingInfo (RenderPBR.psh:299-359): the layers' inputs
    float ccFactor = 0.0f;
    v3    ccN = N;
    SurfaceReflectance ccSrf{};
    if (clearCoat) // ReadClearcoatLayerProperties (:186-220)
    {
        const v4 c = ld<v4>(ly.clearcoat, x, y);
        ccFactor   = c.x;
        ccSrf      = surface_reflectance_clear_coat(c.y, 1.5f);
        if (ly.hasClearcoatNormal) ccN = xyz(ld<v4>(ly.clearcoatNormal, x, y));
    }
    v3    sheenColor = mk3(0.0f);
    float sheenRoughness = 0.0f;
    if (sheen) // ReadSheenLayerProperties (:222-234)
    {
        const v4 c = ld<v4>(ly.sheen, x, y);
        sheenColor = xyz(c);
        sheenRoughness = c.w;
    }
    AnisotropyInfo an{};
    if (aniso) // ReadAnisotropyProperties (:257-297)
    {
        const v4 packed = ld<v4>(ly.anisotropy, x, y);
        const v2 dir{packed.x * ly.rotationCos - packed.y * ly.rotationSin, packed.x * ly.rotationSin + packed.y * ly.rotationCos};
        an.strength = packed.z;
        const v3 T = ly.hasTangent ? xyz(ld<v4>(ly.tangent, x, y)) : v3{1.0f, 0.0f, 0.0f};
        const v3 B = cross(T, N);
        an.tangent   = normalize(dir.x * T + dir.y * B + 0.0f * N); // mul(float3(Direction, 0), MatrixFromRows(Tangent, Bitangent, Normal))
        an.bitangent = cross(N, an.tangent);
        const float pr = srf.perceptualRoughness;
        an.alphaT = lerpf(pr * pr, 1.0f, an.strength * an.strength);
        an.alphaB = pr * pr;
    }
    float iridFactor = 0.0f;
    v3    iridFresnel = mk3(0.0f);
    if (irid) // ReadIridescenceProperties (:236-255) and the blend of :340-347
    {
        const v4 c = ld<v4>(ly.iridescence, x, y);
        iridFactor  = c.x;
        iridFresnel = eval_iridescence(1.0f, ly.iridescenceIor, baseNdotV, c.y, srf.r0);
        const v3 f0 = schlick_to_f0(baseNdotV, iridFresnel, mk3(1.0f));
        if (c.y == 0.0f) iridFactor = 0.0f;
        srf.r0 = lerp3(srf.r0, f0, mk3(iridFactor));
    }
    const float transmission = transm ? ld<float>(ly.transmission, x, y) : 0.0f;

    // ApplyPunctualLight (PBR_Shading.fxh:601-721)
    v3 basePunctualSum = mk3(0.0f), sheenPunctual = mk3(0.0f), ccPunctual = mk3(0.0f);
    const int nl = k.lightCount < MIFX_PBR_MAX_LIGHTS ? k.lightCount : MIFX_PBR_MAX_LIGHTS;
    const BrdfFrame  frame = brdf_frame(N, view, srf), ccFrame = brdf_frame(ccN, view, ccSrf);
    const SheenFrame sheenFrame = sheen ? sheen_frame(sheenRoughness, baseNdotV) : SheenFrame{}; // (dot_sat(N, V) of ApplyDirectionalLightSheen == BaseLayer.NdotV)
    const float albedoScalingV  = sheen ? lut_sample_r(ly.albedoScaling, baseNdotV, sheenRoughness) : 0.0f; // the view half of the albedo scaling (:706-707), once per pixel
    for (int i = 0; i < nl; ++i)
    {
        const mifx_pbr_light_attribs& L = k.lights[i];
        v3    lightDir{L.DirectionX, L.DirectionY, L.DirectionZ};
        float attenuation = 1.0f;
        if (L.Type != MIFX_PBR_LIGHT_TYPE_DIRECTIONAL)
        {
            v3          toPoint = pos - v3{L.PosX, L.PosY, L.PosZ};
            const float d2      = dot(toPoint, toPoint);
            toPoint             = toPoint / fsqrt(d2);
            float rangeAtt      = fdiv(1.0f, d2);
            if (L.Range4 > 0.0f) rangeAtt *= saturate(1.0f - fdiv(d2 * d2, L.Range4));
            if (L.Type == MIFX_PBR_LIGHT_TYPE_POINT) lightDir = toPoint;
            float angular = 1.0f;
            if (L.Type == MIFX_PBR_LIGHT_TYPE_SPOT) angular = saturate(dot(toPoint, lightDir) * L.SpotAngleScale + L.SpotAngleOffset);
            attenuation = rangeAtt * angular;
        }
        if (SHADOWS && L.ShadowMapIndex >= 0) // ENABLE_SHADOWS (:644-660), as apply_punctual_light<true> of pbr.hip
        {
            const mifx_pbr_shadow_map_info& info = sh.info[L.ShadowMapIndex];
            const float* M = info.WorldToLightProjSpace;
            const float w  = pos.x * M[3] + pos.y * M[7] + pos.z * M[11] + M[15];
            const v2 ndc{fdiv(pos.x * M[0] + pos.y * M[4] + pos.z * M[8] + M[12], w), fdiv(pos.x * M[1] + pos.y * M[5] + pos.z * M[9] + M[13], w)};
            const float z  = pos.x * M[2] + pos.y * M[6] + pos.z * M[10] + M[14];
            const v2 t     = ndc_to_uv(ndc);
            const v2 uv{t.x * info.UVScale[0] + info.UVBias[0], t.y * info.UVScale[1] + info.UVBias[1]};
            attenuation *= filter_shadow_map_fixed_pcf(sh, uv, info.ShadowMapSlice, z);
        }
        if (attenuation <= 0.0f) continue;
        const v3 intensity = v3{L.IntensityR, L.IntensityG, L.IntensityB} * attenuation;
        v3    diff, spec;
        float NdotL;
        if (aniso) smith_ggx_brdf_anisotropic(-lightDir, N, view, an, srf, diff, spec, NdotL);
        else smith_ggx_brdf(-lightDir, frame, srf, diff, spec, NdotL);
        if (transm) diff = diff * (1.0f - transmission);
        v3 basePunctual = (diff + spec) * intensity * NdotL;
        if (sheen)
        {
            sheenPunctual += apply_directional_light_sheen(sheenFrame, lightDir, intensity, sheenColor, sheenRoughness, N, view);
            const float maxFactor = fmaxf(fmaxf(sheenColor.x, sheenColor.y), sheenColor.z);
            const float scaling = fminf(1.0f - maxFactor * albedoScalingV, 1.0f - maxFactor * lut_sample_r(ly.albedoScaling, NdotL, sheenRoughness));
            basePunctual = basePunctual * scaling;
        }
        basePunctualSum += basePunctual;
        if (clearCoat) // ApplyDirectionalLightGGX (:122-131)
        {
            v3    d, s;
            float ccNdotL;
            smith_ggx_brdf(-lightDir, ccFrame, ccSrf, d, s, ccNdotL);
            const v3 shade = (d + s) * ccNdotL;
            ccPunctual += intensity * shade;
        }
    }

    // ApplyIBL (:724-792)
    IBLInfo ibl = ibl_sampling_info(srf, lut, N, view);
    if (irid) ibl.kS = lerp3(ibl.kS, iridFresnel, iridFactor);
    v3 diffuseIBL = lambertian_ibl(srf, ibl, xyz(irradianceAt(ibl.N)));
    if (transm) diffuseIBL = diffuseIBL * (1.0f - transmission);
    if (aniso) // the bent normal of KHR_materials_anisotropy (:754-767)
    {
        const v3    anisoTangent = cross(an.bitangent, view);
        const v3    anisoNormal  = cross(anisoTangent, an.bitangent);
        const float bend  = 1.0f - an.strength * (1.0f - srf.perceptualRoughness);
        const float bend4 = bend * bend * bend * bend;
        ibl.N = normalize(lerp3(anisoNormal, N, mk3(bend4)));
        ibl.L = normalize(reflect(-ibl.V, ibl.N));
    }
    const v3 specularIBL = specular_ibl_ggx(ibl, xyz(prefilteredAt(ibl.L, srf.perceptualRoughness * k.prefilteredCubeLastMip)));
    v3 sheenIBL = mk3(0.0f), ccIBL = mk3(0.0f);
    if (sheen) // GetSpecularIBL_Charlie (:347-368)
    {
        const float lod  = sheenRoughness * k.prefilteredCubeLastMip;
        const v3    refl = normalize(reflect(-view, N));
        const float brdf = lut_sample_r(ly.charlie, baseNdotV, sheenRoughness);
        sheenIBL = xyz(prefilteredAt(refl, lod)) * sheenColor * brdf;
    }
    if (clearCoat) // GetClearcoatIBLSamplingInfo (:270-290)
    {
        IBLInfo c;
        c.N = ccN;
        c.V = view;
        c.L = normalize(reflect(-view, ccN));
        c.NdotV  = fmaxf(dot(ccN, view), 0.1f);
        c.preInt = lut_sample(lut, c.NdotV, ccSrf.perceptualRoughness);
        c.kS     = ccSrf.r0;
        ccIBL = specular_ibl_ggx(c, xyz(prefilteredAt(c.L, ccSrf.perceptualRoughness * k.prefilteredCubeLastMip)));
    }

    // ResolveLighting (:847-876)
    v3 color = basePunctualSum + (diffuseIBL + specularIBL) * iblScale * occl + emis;
    if (sheen) color += sheenPunctual + sheenIBL * iblScale * occl;
    if (clearCoat)
    {
        const float ccNdotV = fmaxf(dot(ccN, view), 0.1f);
        const float fresnel = schlick_reflection1(ccNdotV, ccSrf.r0.x, ccSrf.r90.x);
        color = color * (1.0f - ccFactor * fresnel) + (ccPunctual * ccFactor + ccIBL * iblScale * occl * ccFactor);
    }
    outColor = mk4(color, bc.w);
    outSpec  = mk4(specularIBL * iblScale * occl, 1.0f); // GetBaseLayerSpecularIBL (:801-805)
}
} // namespace mifx
