// mifx_pbr.h -- device implementation of the reference's PBR lighting library (metallic-roughness workflow, GGX/Smith
// specular + Lambert diffuse, punctual lights, split-sum IBL with Fdez-Aguera multiple scattering) and of the software
// cube-map / LUT sampling it needs.  Follows Shaders/Common/public/PBR_Common.fxh and Shaders/PBR/public/PBR_Shading.fxh.
#pragma once
#include "mifx.h"
#include "mifx_device.h"

namespace mifx
{
#define MIFX_PI 3.141592653589793f

// ------------------------------------------------------------------------------------------------ PBR_Common.fxh
MIFX_D float dot_sat(v3 a, v3 b) { return saturate(dot(a, b)); }
MIFX_D float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
// SCHLICK_REFLECTION (:81)
MIFX_D v3 schlick_reflection(float VdotH, v3 r0, v3 r90) { return r0 + (r90 - r0) * pow5(clampf(1.0f - VdotH, 0.0f, 1.0f)); }
// SmithGGXVisibilityCorrelated (:107-123)
MIFX_D float smith_ggx_visibility_correlated(float NdotL, float NdotV, float alpha)
{
    const float a2   = alpha * alpha;
    const float ggxv = NdotL * fsqrt(fmaxf(NdotV * NdotV * (1.0f - a2) + a2, 1e-7f));
    const float ggxl = NdotV * fsqrt(fmaxf(NdotL * NdotL * (1.0f - a2) + a2, 1e-7f));
    return fdiv(0.5f, ggxv + ggxl);
}
// the same with the factor that depends on NdotV only hoisted out of a loop over lights / samples:
//   visV = smith_ggx_visibility_v_term(NdotV, alpha);  Vis = smith_ggx_visibility_correlated_v(NdotL, NdotV, alpha, visV)
MIFX_D float smith_ggx_visibility_v_term(float NdotV, float alpha)
{
    const float a2 = alpha * alpha;
    return fsqrt(fmaxf(NdotV * NdotV * (1.0f - a2) + a2, 1e-7f));
}
MIFX_D float smith_ggx_visibility_correlated_v(float NdotL, float NdotV, float alpha, float visV)
{
    const float a2   = alpha * alpha;
    const float ggxv = NdotL * visV;
    const float ggxl = NdotV * fsqrt(fmaxf(NdotL * NdotL * (1.0f - a2) + a2, 1e-7f));
    return fdiv(0.5f, ggxv + ggxl);
}
// SmithGGXMasking (:149-175)
MIFX_D float smith_ggx_masking(float NdotV, float alpha)
{
    const float a2    = alpha * alpha;
    const float denom = NdotV + fsqrt(a2 + (1.0f - a2) * NdotV * NdotV);
    return fdiv(2.0f * fmaxf(NdotV, 0.0f), fmaxf(denom, 1e-6f));
}
// NormalDistribution_GGX (:181-194)
MIFX_D float normal_distribution_ggx(float NdotH, float alpha)
{
    alpha             = fmaxf(alpha, 1e-3f);
    const float a2    = alpha * alpha;
    const float nh2   = NdotH * NdotH;
    const float f     = nh2 * a2 + (1.0f - nh2);
    return fdiv(a2, fmaxf(MIFX_PI * f * f, 1e-9f));
}
// The same three terms with the 1-ulp hardware reciprocal / square root (q_rcp, q_sqrt) in place of the correctly rounded division / square root, for the SSR passes
// that evaluate them per ray or per filter tap (R4, R5; round 3).  Each is a sum or product of non-negative terms followed by one reciprocal: no cancellation, so
// the result moves by ~2e-7 relative -- four orders of magnitude inside the parity contract -- and nothing downstream of them selects texels or crosses a threshold
// other than the 1e-6 floors of the weights.  What feeds them keeps the strict path: NdotH in particular (f = nh2 a2 + (1 - nh2) cancels for smooth surfaces, where one
// ulp of NdotH is 1e-3 of D).  16 + 5 + 16 vector instructions less per evaluation.
MIFX_D float smith_ggx_visibility_correlated_v_q(float NdotL, float NdotV, float alpha, float visV)
{
    const float a2   = alpha * alpha;
    const float ggxv = NdotL * visV;
    const float ggxl = NdotV * q_sqrt(fmaxf(NdotL * NdotL * (1.0f - a2) + a2, 1e-7f));
    return 0.5f * q_rcp(ggxv + ggxl);
}
MIFX_D float smith_ggx_masking_q(float NdotV, float alpha)
{
    const float a2    = alpha * alpha;
    const float denom = NdotV + q_sqrt(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * fmaxf(NdotV, 0.0f) * q_rcp(fmaxf(denom, 1e-6f));
}
MIFX_D float normal_distribution_ggx_q(float NdotH, float alpha)
{
    alpha             = fmaxf(alpha, 1e-3f);
    const float a2    = alpha * alpha;
    const float nh2   = NdotH * NdotH;
    const float f     = nh2 * a2 + (1.0f - nh2);
    return a2 * q_rcp(fmaxf(MIFX_PI * f * f, 1e-9f));
}
// SmithGGXSampleVisibleNormalSC (:278-295)
MIFX_D v3 smith_ggx_sample_visible_normal_sc(v3 view, float ax, float ay, float u1, float u2)
{
    const v3    V   = normalize(view * v3{ax, ay, 1.0f});
    const float phi = 2.0f * MIFX_PI * u1;
    const float z   = (1.0f - u2) * (1.0f + V.z) - V.z;
    const float st  = fsqrt(clampf(1.0f - z * z, 0.0f, 1.0f));
    const v3    H   = v3{st * m_cos(phi), st * m_sin(phi), z} + V;
    return normalize(v3{ax * H.x, ay * H.y, H.z});
}

struct SurfaceReflectance // SurfaceReflectanceInfo (:362-368)
{
    float perceptualRoughness;
    v3    r0, r90, diffuse;
};

// GetSurfaceReflectance, PBR_WORKFLOW_METALLIC_ROUGHNESS branch (PBR_Shading.fxh:376-426)
MIFX_D SurfaceReflectance surface_reflectance_workflow_mr(v3 baseColor, float roughnessG, float metallicB)
{
    SurfaceReflectance s;
    const v3 f0 = mk3(0.04f);
    s.perceptualRoughness = roughnessG;
    s.diffuse             = baseColor * (mk3(1.0f) - f0) * (1.0f - metallicB);
    const v3 spec         = lerp3(f0, baseColor, metallicB);
    s.perceptualRoughness = clampf(s.perceptualRoughness, 0.0f, 1.0f);
    const float r90       = clampf(max_comp(spec) * 50.0f, 0.0f, 1.0f);
    s.r0  = spec;
    s.r90 = mk3(r90);
    return s;
}
// GetPerceivedBrightness / SolveMetallic (PBR_Shading.fxh:93-117)
MIFX_D float perceived_brightness(v3 c) { return fsqrt(0.299f * c.x * c.x + 0.587f * c.y * c.y + 0.114f * c.z * c.z); }
MIFX_D float solve_metallic(v3 diffuse, v3 specular, float oneMinusSpecularStrength)
{
    const float minReflectance = 0.04f;
    const float specularBrightness = perceived_brightness(specular);
    if (specularBrightness < minReflectance) return 0.0f;
    const float diffuseBrightness = perceived_brightness(diffuse);
    const float a = minReflectance;
    const float b = fdiv(diffuseBrightness * oneMinusSpecularStrength, 1.0f - minReflectance) + specularBrightness - 2.0f * minReflectance;
    const float c = minReflectance - specularBrightness;
    const float D = b * b - 4.0f * a * c;
    return clampf(fdiv(-b + fsqrt(D), 2.0f * a), 0.0f, 1.0f);
}
// GetSurfaceReflectance, PBR_WORKFLOW_SPECULAR_GLOSSINESS branch (PBR_Shading.fxh:390-403) on the PhysicalDesc of ReadBaseLayerProperties (RenderPBR.psh:151-165:
// FastSRGBToLinear of the specular colour, SpecularFactor = GlossinessFactor = 1); `metallic` = the value the shader hands to the Material target
MIFX_D SurfaceReflectance surface_reflectance_workflow_sg(v3 baseColor, v4 physicalDesc, float& metallic)
{
    SurfaceReflectance s;
    const v3 f0 = pow3(xyz(physicalDesc), 2.2f); // FastSRGBToLinear (SRGBUtilities.fxh:17-20)
    s.perceptualRoughness = 1.0f - physicalDesc.w;
    const float oneMinusSpecularStrength = 1.0f - max_comp(f0);
    s.diffuse = baseColor * oneMinusSpecularStrength;
    metallic  = solve_metallic(baseColor, f0, oneMinusSpecularStrength);
    s.perceptualRoughness = clampf(s.perceptualRoughness, 0.0f, 1.0f);
    const float r90 = clampf(max_comp(f0) * 50.0f, 0.0f, 1.0f);
    s.r0  = f0;
    s.r90 = mk3(r90);
    return s;
}
// GetSurfaceReflectanceMR (PBR_Shading.fxh:429-449), used by the composite pass
MIFX_D SurfaceReflectance surface_reflectance_mr(v3 baseColor, float metallic, float roughness)
{
    SurfaceReflectance s;
    const float f0 = 0.04f;
    s.perceptualRoughness = roughness;
    s.diffuse             = baseColor * ((1.0f - f0) * (1.0f - metallic));
    const v3 r0           = lerp3(mk3(f0), baseColor, metallic);
    const float r90       = fminf(max_comp(r0) * 50.0f, 1.0f);
    s.r0  = r0;
    s.r90 = mk3(r90);
    return s;
}

// SmithGGX_BRDF (PBR_Common.fxh:371-405) incl. GetAngularInfo (:340-360)
// The light-independent part of GetAngularInfo / SmithGGX_BRDF, evaluated once per pixel instead of once per light (the light loop has an
// early-out, so the compiler does not hoist it by itself): two normalisations, NdotV, alpha, Diffuse / PI -- 64 of ~200 instructions per light.
struct BrdfFrame
{
    v3    n, v, diffuseOverPi;
    float NdotV, alpha, visV;
};
MIFX_D BrdfFrame brdf_frame(v3 normal, v3 view, const SurfaceReflectance& srf)
{
    BrdfFrame f;
    f.n = normalize(normal);
    f.v = normalize(view);
    f.NdotV = dot_sat(f.n, f.v);
    f.alpha = srf.perceptualRoughness * srf.perceptualRoughness;
    f.diffuseOverPi = srf.diffuse / MIFX_PI;
    f.visV = smith_ggx_visibility_v_term(f.NdotV, f.alpha);
    return f;
}
MIFX_D void smith_ggx_brdf(v3 pointToLight, const BrdfFrame& f, const SurfaceReflectance& srf, v3& diffuse, v3& spec, float& NdotL)
{
    const v3 l = normalize(pointToLight), h = normalize(l + f.v);
    NdotL = dot_sat(f.n, l);
    const float NdotH = dot_sat(f.n, h), VdotH = dot_sat(f.v, h);
    diffuse = mk3(0.0f);
    spec    = mk3(0.0f);
    if (NdotL > 0.0f || f.NdotV > 0.0f)
    {
        const float D   = normal_distribution_ggx(NdotH, f.alpha);
        const float Vis = smith_ggx_visibility_correlated_v(NdotL, f.NdotV, f.alpha, f.visV);
        const v3    F   = schlick_reflection(VdotH, srf.r0, srf.r90);
        diffuse = (mk3(1.0f) - F) * f.diffuseOverPi;
        spec    = F * Vis * D;
    }
}

// ------------------------------------------------------------------------------------------------ software texture sampling for the IBL inputs
struct LutK // preintegrated GGX BRDF (PrecomputeBRDF.psh), rg used
{
    const float* data;
    int size_w, size_h, pitch_f, comps;
};
MIFX_D v2 lut_sample(const LutK& t, float u, float v) // .Sample(Sam_LinearClamp) on a 1-mip texture
{
    const Bilinear b = bilinear_uc(u * float(t.size_w), v * float(t.size_h), t.size_w, t.size_h);
    auto ldp = [&](int x, int y) { const float* p = t.data + size_t(y) * t.pitch_f + size_t(x) * t.comps; return v2{p[0], p[1]}; };
    return ldp(b.x0, b.y0) * b.w00 + ldp(b.x1, b.y0) * b.w10 + ldp(b.x0, b.y1) * b.w01 + ldp(b.x1, b.y1) * b.w11;
}

struct CubeK // D3D face order +X,-X,+Y,-Y,+Z,-Z; faces of a mip stacked vertically, float4 texels, tightly packed
{
    const v4* mip[12];
    int size, mips;
};
MIFX_D void cube_face_uv(v3 d, int& face, float& u, float& v)
{
    // D3D major-axis selection, written with selects (the three-level branch version cost more scalar control flow than arithmetic):
    //   +X: sc = -z, tc = -y   -X: sc = z, tc = -y   +Y: sc = x, tc = z   -Y: sc = x, tc = -z   +Z: sc = x, tc = -y   -Z: sc = -x, tc = -y
    const float ax = fabsf(d.x), ay = fabsf(d.y), az = fabsf(d.z);
    const bool  isX = ax >= ay && ax >= az, isY = !isX && ay >= az;
    const bool  px = d.x >= 0.0f, py = d.y >= 0.0f, pz = d.z >= 0.0f;
    const float ma = isX ? ax : (isY ? ay : az);
    face = isX ? (px ? 0 : 1) : (isY ? (py ? 2 : 3) : (pz ? 4 : 5));
    const float sc = isX ? (px ? -d.z : d.z) : (isY ? d.x : (pz ? d.x : -d.x));
    const float tc = isY ? (py ? d.z : -d.z) : -d.y;
    u = 0.5f * (fdiv(sc, ma) + 1.0f);
    v = 0.5f * (fdiv(tc, ma) + 1.0f);
}
MIFX_D v3 cube_dir(int face, float u, float v)
{
    const float sc = 2.0f * u - 1.0f, tc = 2.0f * v - 1.0f;
    //   0: (1, -tc, -sc)  1: (-1, -tc, sc)  2: (sc, 1, tc)  3: (sc, -1, -tc)  4: (sc, -tc, 1)  5: (-sc, -tc, -1)
    const float one = (face & 1) ? -1.0f : 1.0f;
    const int   axis = face >> 1;
    return v3{axis == 0 ? one : (face == 5 ? -sc : sc),
              axis == 1 ? one : -tc,
              axis == 0 ? (face == 0 ? -sc : sc) : (axis == 1 ? (face == 2 ? tc : -tc) : one)};
}
// Bilinear taps outside the face are re-projected onto the cube and resolved to the nearest texel of the face they land on
// (filtering contract shared with the oracle, see oracle/ref/hlsl_shim.h hl_cube_texel).
MIFX_D v4 cube_texel(const v4* im, int n, int face, int x, int y)
{
    if (x < 0 || y < 0 || x >= n || y >= n)
    {
        const v3 d = cube_dir(face, fdiv(float(x) + 0.5f, float(n)), fdiv(float(y) + 0.5f, float(n)));
        float u, v;
        cube_face_uv(d, face, u, v);
        x = clampi(int(floorf(u * float(n))), 0, n - 1);
        y = clampi(int(floorf(v * float(n))), 0, n - 1);
    }
    return im[size_t(face * n + y) * n + x];
}
MIFX_D v4 cube_sample_level(const v4* im, int n, v3 dir)
{
    int face; float u, v;
    cube_face_uv(dir, face, u, v);
    const float fx = u * float(n) - 0.5f, fy = v * float(n) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int   x0 = int(x0f), y0 = int(y0f);
    v4 acc = cube_texel(im, n, face, x0, y0) * ((1.0f - wx) * (1.0f - wy));
    acc += cube_texel(im, n, face, x0 + 1, y0) * (wx * (1.0f - wy));
    acc += cube_texel(im, n, face, x0, y0 + 1) * ((1.0f - wx) * wy);
    acc += cube_texel(im, n, face, x0 + 1, y0 + 1) * (wx * wy);
    return acc;
}
// SampleLevel(Sam_LinearClamp, dir, lod): trilinear.  `mip` is the table of level pointers: the kernel-argument copy for a uniform lod, an LDS
// copy (stage_cube_mips) when the lod differs per lane -- indexing the kernel argument with a VGPR costs a dependent memory round trip per level
MIFX_D v4 cube_sample(const v4* const* mip, int size, int mips, v3 dir, float lod)
{
    const float maxl = float(mips - 1);
    lod = fminf(fmaxf(lod, 0.0f), maxl);
    const int   l0 = int(floorf(lod));
    const int   l1 = l0 + 1 < mips ? l0 + 1 : l0;
    const float f  = lod - float(l0);
    const v4 a = cube_sample_level(mip[l0], size >> l0 > 0 ? size >> l0 : 1, dir);
    if (f == 0.0f || l1 == l0) return a;
    const v4 b = cube_sample_level(mip[l1], size >> l1 > 0 ? size >> l1 : 1, dir);
    return a + (b - a) * f;
}
MIFX_D v4 cube_sample(const CubeK& c, v3 dir, float lod) { return cube_sample(c.mip, c.size, c.mips, dir, lod); }

// Apron layout: every face of a level carries a one-texel border, (n + 2) x (n + 2) texels per face; the border texel (x, y), x or y in {-1, n},
// holds cube_texel(face, x, y), i.e. the texel the filtering contract resolves an outside tap to.  A bilinear footprint then never leaves its
// face block and sampling needs no face-edge branch (in the plain layout nearly every wave has some lane on a face edge and executes the
// re-projection path for each of its 12 taps).  Built per shading call by cube_apron_kernel (pbr.hip).
MIFX_D v4 cube_sample_level_apron(const v4* im, int n, v3 dir)
{
    int face; float u, v;
    cube_face_uv(dir, face, u, v);
    const float fx = u * float(n) - 0.5f, fy = v * float(n) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int   m = n + 2;
    const v4*   p = im + size_t(face * m + int(y0f) + 1) * m + (int(x0f) + 1); // x0, y0 in [-1, n - 1]
    v4 acc = p[0] * ((1.0f - wx) * (1.0f - wy));
    acc += p[1] * (wx * (1.0f - wy));
    acc += p[m] * ((1.0f - wx) * wy);
    acc += p[m + 1] * (wx * wy);
    return acc;
}
MIFX_D v4 cube_sample_apron(const v4* const* mip, int size, int mips, v3 dir, float lod) // same level selection and blend as cube_sample
{
    const float maxl = float(mips - 1);
    lod = fminf(fmaxf(lod, 0.0f), maxl);
    const int   l0 = int(floorf(lod));
    const int   l1 = l0 + 1 < mips ? l0 + 1 : l0;
    const float f  = lod - float(l0);
    const v4 a = cube_sample_level_apron(mip[l0], size >> l0 > 0 ? size >> l0 : 1, dir);
    if (f == 0.0f || l1 == l0) return a;
    const v4 b = cube_sample_level_apron(mip[l1], size >> l1 > 0 ? size >> l1 : 1, dir);
    return a + (b - a) * f;
}
MIFX_D void stage_cube_mips(const v4** lds, const CubeK& c) // call from every thread of the block before any early return
{
    const unsigned t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < 12u) lds[t] = c.mip[t];
    __syncthreads();
}

// ------------------------------------------------------------------------------------------------ IBL (PBR_Shading.fxh:220-345)
struct IBLInfo
{
    v3    N, V, L;
    float NdotV;
    v2    preInt;
    v3    kS;
};
MIFX_D IBLInfo ibl_sampling_info(const SurfaceReflectance& srf, const LutK& lut, v3 N, v3 V) // GetIBLSamplingInfo :232-268
{
    IBLInfo i;
    i.N = N;
    i.V = V;
    i.L = normalize(reflect(-V, N));
    i.NdotV  = dot_sat(N, V);
    i.preInt = lut_sample(lut, i.NdotV, srf.perceptualRoughness);
    const float omr = 1.0f - srf.perceptualRoughness;
    const v3    r90 = max3(mk3(omr), srf.r0);
    i.kS = schlick_reflection(i.NdotV, srf.r0, r90);
    return i;
}
MIFX_D v3 specular_ibl_ggx(const IBLInfo& i, v3 specularLight) { return specularLight * (i.kS * i.preInt.x + i.preInt.y); } // :293-304
MIFX_D v3 lambertian_ibl(const SurfaceReflectance& srf, const IBLInfo& i, v3 irradiance)                                    // :317-345
{
    const v3    FssEss = i.kS * i.preInt.x + i.preInt.y;
    const float Ess    = i.preInt.x + i.preInt.y;
    const float Ems    = 1.0f - Ess;
    const v3    Favg   = srf.r0 + (mk3(1.0f) - srf.r0) / 21.0f;
    const v3    Fms    = FssEss * Favg / (mk3(1.0f) - Ems * Favg);
    const v3    Edss   = mk3(1.0f) - (FssEss + Fms * Ems);
    const v3    kD     = srf.diffuse * Edss;
    return (Fms * Ems + kD) * irradiance;
}

// ------------------------------------------------------------------------------------------------ shadow-mapped punctual lights (used by pbr.hip and mifx_pbr_layers.h)
// ---- shadow map of the punctual lights (ENABLE_SHADOWS, RenderPBR.psh:70-73): Texture2DArray<float> sampled with Sam_ComparisonLinearClamp
struct ShadowK
{
    const unsigned char* data;
    int                  w, h, slices, pitch;
    unsigned long long   slicePitch;
    int                  pcf; // PCF_FILTER_SIZE
    mifx_pbr_shadow_map_info info[MIFX_PBR_MAX_SHADOW_MAPS];
};
// SampleCmpLevelZero: the bilinear blend of "reference < texel" over the clamped 2x2 footprint of the slice nearest to `slice`
MIFX_D float sample_cmp_level_zero(const ShadowK& sh, float u, float v, float slice, float ref)
{
    const int   s    = clampi(int(floorf(slice + 0.5f)), 0, sh.slices - 1);
    const Img   im{const_cast<unsigned char*>(sh.data) + size_t(s) * sh.slicePitch, sh.w, sh.h, sh.pitch, 0, 0};
    const Bilinear b = bilinear_uc(u * float(sh.w), v * float(sh.h), sh.w, sh.h);
    const float c00 = ref < ld<float>(im, b.x0, b.y0) ? 1.0f : 0.0f, c10 = ref < ld<float>(im, b.x1, b.y0) ? 1.0f : 0.0f;
    const float c01 = ref < ld<float>(im, b.x0, b.y1) ? 1.0f : 0.0f, c11 = ref < ld<float>(im, b.x1, b.y1) ? 1.0f : 0.0f;
    return c00 * b.w00 + c10 * b.w10 + c01 * b.w01 + c11 * b.w11;
}
// FilterShadowMapFixedPCF (Shaders/Common/public/PCF.fxh:7-152; "the method used in The Witness"), receiver-plane depth bias (0, 0) as ApplyPunctualLight passes it
MIFX_D float filter_shadow_map_fixed_pcf(const ShadowK& sh, v2 uvIn, float slice, float depth)
{
    const float sx = float(sh.w), sy = float(sh.h), isx = fdiv(1.0f, sx), isy = fdiv(1.0f, sy);
    const v2    uv{uvIn.x * sx, uvIn.y * sy};
    v2          base{floorf(uv.x + 0.5f), floorf(uv.y + 0.5f)};
    const float s = uv.x + 0.5f - base.x, t = uv.y + 0.5f - base.y;
    base = v2{(base.x - 0.5f) * isx, (base.y - 0.5f) * isy};
    const float ref = fmaxf(depth, 1e-8f); // DepthClamp
    auto S = [&](float u, float v) { return sample_cmp_level_zero(sh, base.x + u * isx, base.y + v * isy, slice, ref); };
    float sum = 0.0f;
    if (sh.pcf == 2) return sample_cmp_level_zero(sh, uvIn.x, uvIn.y, slice, ref);
    if (sh.pcf == 3)
    {
        const float uw0 = 3.0f - 2.0f * s, uw1 = 1.0f + 2.0f * s, u0 = fdiv(2.0f - s, uw0) - 1.0f, u1 = fdiv(s, uw1) + 1.0f;
        const float vw0 = 3.0f - 2.0f * t, vw1 = 1.0f + 2.0f * t, v0 = fdiv(2.0f - t, vw0) - 1.0f, v1 = fdiv(t, vw1) + 1.0f;
        sum += uw0 * vw0 * S(u0, v0); sum += uw1 * vw0 * S(u1, v0); sum += uw0 * vw1 * S(u0, v1); sum += uw1 * vw1 * S(u1, v1);
        return fdiv(sum * 1.0f, 16.0f);
    }
    if (sh.pcf == 5)
    {
        const float uw[3] = {4.0f - 3.0f * s, 7.0f, 1.0f + 3.0f * s}, vw[3] = {4.0f - 3.0f * t, 7.0f, 1.0f + 3.0f * t};
        const float u[3] = {fdiv(3.0f - 2.0f * s, uw[0]) - 2.0f, fdiv(3.0f + s, uw[1]), fdiv(s, uw[2]) + 2.0f};
        const float v[3] = {fdiv(3.0f - 2.0f * t, vw[0]) - 2.0f, fdiv(3.0f + t, vw[1]), fdiv(t, vw[2]) + 2.0f};
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i) sum += uw[i] * vw[j] * S(u[i], v[j]);
        return fdiv(sum * 1.0f, 144.0f);
    }
    if (sh.pcf == 7)
    {
        const float uw[4] = {5.0f * s - 6.0f, 11.0f * s - 28.0f, -(11.0f * s + 17.0f), -(5.0f * s + 1.0f)}, vw[4] = {5.0f * t - 6.0f, 11.0f * t - 28.0f, -(11.0f * t + 17.0f), -(5.0f * t + 1.0f)};
        const float u[4] = {fdiv(4.0f * s - 5.0f, uw[0]) - 3.0f, fdiv(4.0f * s - 16.0f, uw[1]) - 1.0f, fdiv(-(7.0f * s + 5.0f), uw[2]) + 1.0f, fdiv(-s, uw[3]) + 3.0f};
        const float v[4] = {fdiv(4.0f * t - 5.0f, vw[0]) - 3.0f, fdiv(4.0f * t - 16.0f, vw[1]) - 1.0f, fdiv(-(7.0f * t + 5.0f), vw[2]) + 1.0f, fdiv(-t, vw[3]) + 3.0f};
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < 4; ++i) sum += uw[i] * vw[j] * S(u[i], v[j]);
        return fdiv(sum * 1.0f, 2704.0f);
    }
    return 0.0f;
}

// ------------------------------------------------------------------------------------------------ the constant block of the shade kernels (pbr.hip)
struct ShadeK
{
    float iblScale[3];
    float occlusionStrength, emissionScale, prefilteredCubeLastMip;
    int   lightCount;
    mifx_pbr_light_attribs lights[MIFX_PBR_MAX_LIGHTS];
    float background[4];
    int   workflow; // MIFX_PBR_WORKFLOW_*
};

} // namespace mifx
