// pbr.hip -- (P1-P9) per-pixel PBR shade from a G-buffer (the SSR / SSAO composite M1 is in composite.hip).
//
// P*: the lighting half of Shaders/PBR/private/RenderPBR.psh (GetSurfaceShadingInfo :299-359 -> ApplyPunctualLight x N :479-499 ->
//     ApplyIBL :501-512 -> ResolveLighting :514), material fetch replaced by the G-buffer (contract: PBR/src/USD_Renderer.cpp:83-162),
//     world position rebuilt from depth with InvProjectPosition (PostFX_Common.fxh:99-105).  84 B/px: base colour, normal, material
//     (3 x 16 B) + depth (4 B) in, radiance + specular IBL (2 x 16 B) out; LUT and cube maps are cache-resident (<= 9 MB).
#include "mifx_host.h"
#include <cmath>
#include <cstdlib>
#include "mifx_pbr.h"
#include "mifx_pbr_layers.h"
#include "mifx_effects.h"
#include "mifx_tonemap.h"
#include "mifx_formats.h"

namespace mifx
{
// ApplyPunctualLight (PBR_Shading.fxh:601-721); sheen / clear coat / anisotropy compiled out (defaults, PBR_Renderer.hpp:159-179); SHADOWS = ENABLE_SHADOWS
template <bool SHADOWS>
MIFX_D void apply_punctual_light(v3 pos, const BrdfFrame& frame, const SurfaceReflectance& srf, const mifx_pbr_light_attribs& L, v3& punctual, const ShadowK* sh)
{
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
    if (SHADOWS && L.ShadowMapIndex >= 0) // :644-660
    {
        const mifx_pbr_shadow_map_info& info = sh->info[L.ShadowMapIndex];
        const float* M = info.WorldToLightProjSpace;
        const float w  = pos.x * M[3] + pos.y * M[7] + pos.z * M[11] + M[15];
        const v2 ndc{fdiv(pos.x * M[0] + pos.y * M[4] + pos.z * M[8] + M[12], w), fdiv(pos.x * M[1] + pos.y * M[5] + pos.z * M[9] + M[13], w)};
        const float z  = pos.x * M[2] + pos.y * M[6] + pos.z * M[10] + M[14]; // (not divided by w in the reference)
        const v2 t     = ndc_to_uv(ndc);
        const v2 uv{t.x * info.UVScale[0] + info.UVBias[0], t.y * info.UVScale[1] + info.UVBias[1]};
        attenuation *= filter_shadow_map_fixed_pcf(*sh, uv, info.ShadowMapSlice, z);
    }
    if (attenuation <= 0.0f) return;
    const v3 intensity = v3{L.IntensityR, L.IntensityG, L.IntensityB} * attenuation;
    v3    diff, spec;
    float NdotL;
    smith_ggx_brdf(-lightDir, frame, srf, diff, spec, NdotL);
    punctual += (diff + spec) * intensity * NdotL;
}

// ------------------------------------------------------------------------------------------------ apron copy of the IBL cube maps
struct ApronPlan // levels of one cube laid out back to back in the scratch block
{
    v4* mip[12];
    int first[13]; // first[l] = index of the first texel of level l in the flattened job list; first[levels] = total
    int size, levels;
};
MIFX_D void cube_apron_texel(const CubeK& src, const ApronPlan& plan, int idx)
{
    if (idx >= plan.first[plan.levels]) return;
    int l = 0;
    while (idx >= plan.first[l + 1]) ++l;
    const int n = plan.size >> l > 0 ? plan.size >> l : 1, m = n + 2;
    const int local = idx - plan.first[l];
    const int face = local / (m * m), rem = local - face * m * m;
    const int y = rem / m - 1, x = rem - (rem / m) * m - 1;
    plan.mip[l][local] = cube_texel(src.mip[l], n, face, x, y); // interior: the texel itself; border: the re-projected nearest texel
}
// both cubes of a shading call in one launch (blocks [0, blocksA) copy the first): a launch of a few thousand texels is all fixed cost
__global__ __launch_bounds__(256) void cube_apron_kernel(CubeK srcA, ApronPlan planA, int blocksA, CubeK srcB, ApronPlan planB)
{
    if (int(blockIdx.x) < blocksA) cube_apron_texel(srcA, planA, int(blockIdx.x * blockDim.x + threadIdx.x));
    else cube_apron_texel(srcB, planB, int((blockIdx.x - unsigned(blocksA)) * blockDim.x + threadIdx.x));
}
static size_t apron_bytes(int size, int levels)
{
    size_t t = 0;
    for (int l = 0; l < levels; ++l) { const size_t m = size_t(size >> l > 0 ? size >> l : 1) + 2; t += 6 * m * m * sizeof(v4); }
    return t;
}
static int plan_cube_apron(const CubeK& src, int levels, unsigned char* scratch, CubeK& out, ApronPlan& plan) // returns the number of 256-thread blocks
{
    plan = ApronPlan{};
    plan.size = src.size; plan.levels = levels;
    out = src;
    out.mips = src.mips;
    size_t off = 0;
    int total = 0;
    for (int l = 0; l < 12; ++l) out.mip[l] = nullptr;
    for (int l = 0; l < levels; ++l)
    {
        const int m = (src.size >> l > 0 ? src.size >> l : 1) + 2;
        plan.mip[l] = reinterpret_cast<v4*>(scratch + off);
        out.mip[l]  = plan.mip[l];
        plan.first[l] = total;
        total += 6 * m * m;
        off += size_t(6) * m * m * sizeof(v4);
    }
    plan.first[levels] = total;
    return (total + 255) / 256;
}

// G-buffer access of the shade: fp32 planes (the contract) or the reference's own texture formats (NativeImg, mifx_formats.h) -- the arithmetic between the
// loads and the stores is the same code
MIFX_D bool  px_xy(const Img& o, int& x, int& y) { return pixel_xy(o, x, y); }
MIFX_D float px_f(const Img& i, int x, int y) { return ld<float>(i, x, y); }
MIFX_D v4    px_v4(const Img& i, int x, int y) { return ld<v4>(i, x, y); }
// the G-buffer texel of the thread's OWN pixel: read by this thread only in the whole launch (ld_once).  (The depth, px_f, stays a plain load: the passes behind the shade read
// that plane next.  The hit fetch of row-band sharding shades pixels scattered over the frame -- neighbouring rays end on neighbouring texels, whose lines are asked for again:
// its loads stay plain, OwnPixel<HitOut>; with the hint the pass takes 67.5 instead of 61.1 us per band.)
MIFX_D v4    px_v4_own(const Img& i, int x, int y) { return ld_once<v4>(i, x, y); } // (each G-buffer texel is read by its own pixel only; the depth, px_f, stays a plain load: the passes behind the shade read that plane next)
MIFX_D void  px_st(const Img& i, int x, int y, v4 c) { st_v4_late<2>(i, x, y, c); }
MIFX_D bool  px_xy(const NativeImg& o, int& x, int& y)
{
    x = int(blockIdx.x * blockDim.x + threadIdx.x);
    y = int(blockIdx.y * blockDim.y + threadIdx.y);
    return x < o.w && y < o.h;
}
MIFX_D v4    px_v4(const NativeImg& i, int x, int y) { return decode_texel(i.p + size_t(y) * i.pitch + size_t(x) * i.texel, i.fmt); }
MIFX_D v4    px_v4_own(const NativeImg& i, int x, int y) { return px_v4(i, x, y); }
MIFX_D float px_f(const NativeImg& i, int x, int y) { return px_v4(i, x, y).x; }
MIFX_D void  px_st(const NativeImg& i, int x, int y, v4 c) { encode_texel(i.p + size_t(y) * i.pitch + size_t(x) * i.texel, i.fmt, c); }

// The output of the hit fetch of row-band sharding (launch_pbr_hit_fetch): the lane's "pixel" is the hit of the ray of texel (tx, ty) of `rays`; px_xy() serves the
// hits whose colour this rank holds itself and returns true -- shade this pixel -- for the others.
// (Round 5: R4 now loads the hits in the rank's own rows itself, so the records left here are the pixels to shade.  Measured and not taken: compacting those records
//  inside the workgroup first -- an LDS list filled through an atomic counter, thread i shades entry i, the waves beyond the list leave -- on the assumption that the
//  pass' 68 us per band at 8K / 8 ranks were scattered lanes: 68.6 us with the list against 67.5 us without.  The time is the shading itself: for a band of reflective
//  ground most rays end on geometry above the band, i.e. in rows another rank shaded.)
struct HitOut
{
    Img rays, coords, radiance;
    int shadedBegin, shadedEnd;
};
MIFX_D bool px_xy(const HitOut& o, int& x, int& y)
{
    int tx, ty;
    if (!pixel_xy(o.rays, tx, ty)) return false;
    const unsigned c = __float_as_uint(ld<float>(o.coords, tx, ty));
    if (c == 0xffffffffu) return false; // no ray, no hit, or a hit outside the frame: xyz is 0 already, as in the unsharded kernel
    x = int(c & 0xffffu);
    y = int(c >> 16);
    if (y < o.shadedBegin || y >= o.shadedEnd) return true;
    const v4 r = ld<v4>(o.radiance, x, y);
    st<v4>(o.rays, tx, ty, v4{r.x, r.y, r.z, ld<v4>(o.rays, tx, ty).w});
    return false;
}
template <class OUT> struct OwnPixel { static constexpr bool value = true; };
template <> struct OwnPixel<HitOut> { static constexpr bool value = false; };
MIFX_D void px_st(const HitOut& o, int, int, v4 c)
{
    int tx, ty;
    (void)pixel_xy(o.rays, tx, ty);
    st<v4>(o.rays, tx, ty, v4{c.x, c.y, c.z, ld<v4>(o.rays, tx, ty).w});
}

// irradiance / prefiltered: apron copies (cube_apron_kernel); the prefiltered lod follows the per-pixel roughness
template <bool HAS_EMISSIVE, bool HAS_AO, bool WRITE_SPEC, bool SHADOWS, class IMG, class OUT>
MIFX_D void pbr_shade_body(const IMG& baseColor, const IMG& normalTex, const IMG& material, const IMG& depthTex, const IMG& emissive, const IMG& occlusion, const LutK& lut,
                           const CubeK& irradiance, const CubeK& prefiltered, const OUT& outRadiance, const OUT& outSpecIBL, const CamK& cam, const ShadeK& k, const ShadowK* sh,
                           const SsrMaskOut* r2 = nullptr)
{
    __shared__ const v4* prefMips[12]; // the prefiltered-environment lod follows the per-pixel roughness
    // Round 5 (tools/isa_roundtrips.py): the kernel began with the staging of the mip table -- a vector load from the kernel arguments, a wait, a barrier -- then asked
    // for the depth, then for the material, stored R2's two values and WAITED FOR THE STORES (vmcnt counts them) before it asked for the colour and the normal: four
    // dependent round trips in front of the shading.  Now the depth and the material are requested before the table is staged (one round trip for the three), R2's
    // stores leave with the kernel's own at the end, and the material texel is fetched once for R2 and the shading.  Same loads otherwise, same arithmetic.
    int x = 0, y = 0;
    const bool active = px_xy(outRadiance, x, y);
    const bool doR2   = r2 != nullptr && r2->enabled;
    const int  lx = active ? x : 0, ly = active ? y : 0; // (threads outside the image / without a pixel request texel (0, 0) and drop it)
    float depth = px_f(depthTex, lx, ly);
    v4    m     = OwnPixel<OUT>::value ? px_v4_own(material, lx, ly) : px_v4(material, lx, ly);
    stage_cube_mips(prefMips, prefiltered);
    if (!active) return;
    float r2Rough = 0.0f, r2Mask = 0.0f;
    if (doR2)
    {
        // R2 of ScreenSpaceReflection (SSR_ComputeStencilMaskAndExtractRoughness.fx:13-40) on the material / depth texels this kernel reads anyway: the arithmetic of
        // ssr_mask_roughness_kernel (a channel select and, for squared roughness, the correctly rounded square root: bit-identical), one pass over the frame less
        const v4 sel{r2->channel == 0u ? 1.0f : 0.0f, r2->channel == 1u ? 1.0f : 0.0f, r2->channel == 2u ? 1.0f : 0.0f, r2->channel == 3u ? 1.0f : 0.0f};
        float r = dot(m, sel);
        if (!r2->perceptual) r = fsqrt(r);
        r2Rough = r;
        r2Mask  = is_reflection_sample(r, depth, r2->threshold, cam.reversedDepth != 0) ? 1.0f : 0.0f;
    }
    auto store_r2 = [&]() {
        if (!doR2) return;
        st<rough_t>(r2->roughness, x, y, r2Rough);
        st<mask_t>(r2->mask, x, y, r2Mask);
    };
    if (is_background(depth, cam.reversedDepth != 0))
    {
        px_st(outRadiance, x, y, v4{k.background[0], k.background[1], k.background[2], k.background[3]});
        if (WRITE_SPEC) px_st(outSpecIBL, x, y, mk4(0.0f));
        store_r2();
        return;
    }
    const v4 bc  = OwnPixel<OUT>::value ? px_v4_own(baseColor, x, y) : px_v4(baseColor, x, y);
    const v4 mat = m;
    const v3 N   = xyz(OwnPixel<OUT>::value ? px_v4_own(normalTex, x, y) : px_v4(normalTex, x, y));

    const v3 pos  = inv_project_position(v3{(float(x) + 0.5f) * cam.ivw, (float(y) + 0.5f) * cam.ivh, depth}, cam.viewProjInv);
    const v3 view = normalize(v3{cam.pos[0], cam.pos[1], cam.pos[2]} - pos);
    // ReadBaseLayerProperties (RenderPBR.psh:138-184): metallic-roughness, RoughnessFactor = MetallicFactor = 1
    float unusedMetallic;
    const SurfaceReflectance srf = k.workflow == MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS ? surface_reflectance_workflow_sg(xyz(bc), mat, unusedMetallic)
                                                                                       : surface_reflectance_workflow_mr(xyz(bc), saturate(mat.x * 1.0f), saturate(mat.y * 1.0f));
    float occl = HAS_AO ? px_f(occlusion, x, y) : 1.0f;
    v3    emis = HAS_EMISSIVE ? xyz(px_v4(emissive, x, y)) : mk3(0.0f);
    occl = lerpf(1.0f, occl, k.occlusionStrength);
    emis = emis * k.emissionScale;
    const v3 iblScale{k.iblScale[0], k.iblScale[1], k.iblScale[2]};

    v3 punctual = mk3(0.0f);
    const int nl = k.lightCount < MIFX_PBR_MAX_LIGHTS ? k.lightCount : MIFX_PBR_MAX_LIGHTS;
    const BrdfFrame frame = brdf_frame(N, view, srf);
    for (int i = 0; i < nl; ++i) apply_punctual_light<SHADOWS>(pos, frame, srf, k.lights[i], punctual, sh);

    // ApplyIBL (PBR_Shading.fxh:724-792)
    const IBLInfo ibl = ibl_sampling_info(srf, lut, N, view);
    const v3 diffuseIBL  = lambertian_ibl(srf, ibl, xyz(cube_sample_level_apron(irradiance.mip[0], irradiance.size, ibl.N)));
    const v3 specularIBL = specular_ibl_ggx(ibl, xyz(cube_sample_apron(prefMips, prefiltered.size, prefiltered.mips, ibl.L, srf.perceptualRoughness * k.prefilteredCubeLastMip)));

    // ResolveLighting (:847-876): Punctual + (DiffuseIBL + SpecularIBL) * IBLScale * Occlusion + Emissive
    const v3 color = punctual + (diffuseIBL + specularIBL) * iblScale * occl + emis;
    px_st(outRadiance, x, y, mk4(color, bc.w));
    if (WRITE_SPEC) px_st(outSpecIBL, x, y, mk4(specularIBL * iblScale * occl, 1.0f)); // GetBaseLayerSpecularIBL (:801-805)
    store_r2();
}
template <bool HAS_EMISSIVE, bool HAS_AO, bool WRITE_SPEC>
#ifndef MIFX_PBR_WAVES
#define MIFX_PBR_WAVES 0
#endif
__global__ __launch_bounds__(256) MIFX_WAVES_OPT(MIFX_PBR_WAVES) void pbr_shade_kernel(Img baseColor, Img normalTex, Img material, Img depthTex, Img emissive, Img occlusion, LutK lut, CubeK irradiance,
                                                        CubeK prefiltered, Img outRadiance, Img outSpecIBL, CamK cam, ShadeK k, SsrMaskOut r2)
{
    pbr_shade_body<HAS_EMISSIVE, HAS_AO, WRITE_SPEC, false>(baseColor, normalTex, material, depthTex, emissive, occlusion, lut, irradiance, prefiltered, outRadiance, outSpecIBL, cam, k, nullptr,
                                                            &r2);
}
// the hit fetch of row-band sharding: the same body on the pixels px_xy(HitOut) selects
template <bool HAS_EMISSIVE, bool HAS_AO>
__global__ __launch_bounds__(256) void pbr_hit_fetch_kernel(Img baseColor, Img normalTex, Img material, Img depthTex, Img emissive, Img occlusion, LutK lut, CubeK irradiance, CubeK prefiltered,
                                                            HitOut out, CamK cam, ShadeK k)
{
    pbr_shade_body<HAS_EMISSIVE, HAS_AO, false, false>(baseColor, normalTex, material, depthTex, emissive, occlusion, lut, irradiance, prefiltered, out, out, cam, k, nullptr);
}
// the ENABLE_SHADOWS permutation: same body, lights with a shadow map are attenuated by the PCF filter
template <bool HAS_EMISSIVE, bool HAS_AO, bool WRITE_SPEC>
__global__ __launch_bounds__(256) void pbr_shade_shadowed_kernel(Img baseColor, Img normalTex, Img material, Img depthTex, Img emissive, Img occlusion, LutK lut, CubeK irradiance,
                                                                 CubeK prefiltered, Img outRadiance, Img outSpecIBL, CamK cam, ShadeK k, ShadowK sh)
{
    pbr_shade_body<HAS_EMISSIVE, HAS_AO, WRITE_SPEC, true>(baseColor, normalTex, material, depthTex, emissive, occlusion, lut, irradiance, prefiltered, outRadiance, outSpecIBL, cam, k, &sh);
}
// the G-buffer in the reference's own texture formats (HnBeginFrameTask.cpp:63-69), radiance / IBL targets likewise (RGBA16_FLOAT there).
// HYDROGENT: the formats are exactly Hydrogent's (BaseColor RGBA8_UNORM, Normal RGBA16_FLOAT, Material RG8_UNORM, depth R32_FLOAT, SceneColor / IBL RGBA16_FLOAT):
// the launcher checked that, and writing the constants into the (by-value) descriptors lets the compiler fold every per-load format switch of decode_texel /
// encode_texel away -- the per-format instance of the kernel (1 228 instructions against 7 174).  Measured at 3840x2160 (tools/native_shade_timing.py, round 2):
// 181 us against 175 us for the generic instance -- the format of a plane is uniform, so the switches are scalar branches that cost nothing per pixel.  The generic
// instance therefore stays the default; MIFX_NATIVE_SHADE_PER_FORMAT=1 selects this one (A/B, tests).
template <bool HAS_EMISSIVE, bool HAS_AO, bool WRITE_SPEC, bool HYDROGENT>
__global__ __launch_bounds__(256) void pbr_shade_native_kernel(NativeImg baseColor, NativeImg normalTex, NativeImg material, NativeImg depthTex, NativeImg emissive, NativeImg occlusion,
                                                               LutK lut, CubeK irradiance, CubeK prefiltered, NativeImg outRadiance, NativeImg outSpecIBL, CamK cam, ShadeK k)
{
    if (HYDROGENT)
    {
        baseColor.fmt = MIFX_NATIVE_FORMAT_RGBA8_UNORM;     baseColor.texel = 4u;
        normalTex.fmt = MIFX_NATIVE_FORMAT_RGBA16_FLOAT;    normalTex.texel = 8u;
        material.fmt  = MIFX_NATIVE_FORMAT_RG8_UNORM;       material.texel = 2u;
        depthTex.fmt  = MIFX_NATIVE_FORMAT_R32_FLOAT;       depthTex.texel = 4u;
        outRadiance.fmt = MIFX_NATIVE_FORMAT_RGBA16_FLOAT;  outRadiance.texel = 8u;
        outSpecIBL.fmt  = MIFX_NATIVE_FORMAT_RGBA16_FLOAT;  outSpecIBL.texel = 8u;
    }
    pbr_shade_body<HAS_EMISSIVE, HAS_AO, WRITE_SPEC, false>(baseColor, normalTex, material, depthTex, emissive, occlusion, lut, irradiance, prefiltered, outRadiance, outSpecIBL, cam, k, nullptr);
}

// ------------------------------------------------------------------------------------------------ the shade with material layers (round 4; ENABLE_CLEAR_COAT / SHEEN / ANISOTROPY /
// IRIDESCENCE / TRANSMISSION of PBR_Shading.fxh, a PSO permutation per set in the reference: PBR_Renderer.cpp:1511-1516).  One kernel, the set is a uniform run-time mask:
// a layer that is off takes the code path of the permutation without it (not "the layer with factor 0").  Not the timed path -- see mifx_pbr_layers.h.
// SET: see pbr_shade_layers_pixel.  Occupancy hints were measured and not taken (profiles/r04_layers_timing_waves.txt: 3 waves per SIMD = 168 registers + 108 bytes of scratch,
// -9 % on all five layers; 4 waves = 128 registers + 252 bytes of scratch, +30 %): the kernel is bound by its arithmetic, not by latency.
template <unsigned SET, bool SHADOWS>
__global__ __launch_bounds__(256) void pbr_shade_layers_kernel(Img baseColor, Img normalTex, Img material, Img depthTex, Img emissive, Img occlusion, LutK lut, CubeK irradiance,
                                                               CubeK prefiltered, Img outRadiance, Img outSpecIBL, CamK cam, ShadeK k, LayersK ly, int hasEmissive, int hasAo, int writeSpec, ShadowK sh)
{
    __shared__ const v4* prefMips[12];
    stage_cube_mips(prefMips, prefiltered);
    int x, y;
    if (!pixel_xy(outRadiance, x, y)) return;
    v4 color, spec;
    pbr_shade_layers_pixel<true, SET, SHADOWS>(x, y, baseColor, normalTex, material, depthTex, emissive, occlusion, lut, irradiance.mip[0], irradiance.size, prefMips, prefiltered.size,
                                               prefiltered.mips, cam, k, ly, hasEmissive, hasAo, sh, color, spec);
    st<v4>(outRadiance, x, y, color);
    if (writeSpec) st<v4>(outSpecIBL, x, y, spec);
}
// the hit fetch of row-band sharding for a frame shaded with layers (pbr_hit_fetch_kernel's counterpart: the same pixels, the layered body)
template <bool SHADOWS>
__global__ __launch_bounds__(256) void pbr_hit_fetch_layers_kernel(Img baseColor, Img normalTex, Img material, Img depthTex, Img emissive, Img occlusion, LutK lut, CubeK irradiance,
                                                                   CubeK prefiltered, HitOut out, CamK cam, ShadeK k, LayersK ly, int hasEmissive, int hasAo, ShadowK sh)
{
    __shared__ const v4* prefMips[12];
    stage_cube_mips(prefMips, prefiltered);
    int x, y;
    if (!px_xy(out, x, y)) return;
    v4 color, spec;
    pbr_shade_layers_pixel<true, kLayersRuntime, SHADOWS>(x, y, baseColor, normalTex, material, depthTex, emissive, occlusion, lut, irradiance.mip[0], irradiance.size, prefMips,
                                                          prefiltered.size, prefiltered.mips, cam, k, ly, hasEmissive, hasAo, sh, color, spec);
    px_st(out, x, y, color);
}

static mifx_status make_cubek(const mifx_cubemap* c, const char* what, CubeK& k)
{
    MIFX_REQUIRE(c != nullptr && c->size > 0 && c->mip_count > 0 && c->mip_count <= 12, "%s: bad cube map", what);
    MIFX_REQUIRE((c->size >> (c->mip_count - 1)) >= 1, "%s: mip_count %u too large for size %u", what, c->mip_count, c->size);
    k.size = int(c->size);
    k.mips = int(c->mip_count);
    for (uint32_t i = 0; i < 12; ++i) k.mip[i] = nullptr;
    for (uint32_t i = 0; i < c->mip_count; ++i)
    {
        MIFX_REQUIRE(c->mip_data[i] != nullptr, "%s: mip %u is null", what, i);
        k.mip[i] = static_cast<const v4*>(c->mip_data[i]);
    }
    return MIFX_OK;
}
// (also used by the composite: composite.hip)
mifx_status make_lutk(const mifx_image2d* im, LutK& k)
{
    MIFX_REQUIRE(im != nullptr && im->data != nullptr && (im->format == MIFX_FORMAT_F32X2 || im->format == MIFX_FORMAT_F32X4), "brdf_lut: F32X2 or F32X4 image required");
    k.data   = static_cast<const float*>(im->data);
    k.size_w = int(im->width);
    k.size_h = int(im->height);
    k.comps  = im->format == MIFX_FORMAT_F32X2 ? 2 : 4;
    k.pitch_f = int(im->pitch_bytes / 4u);
    return MIFX_OK;
}

// what both shade launchers share: the IBL look-ups (with their per-call apron copies) and the constant block
static mifx_status make_shade_constants(hipStream_t s, IblApronCache& cache, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], LutK& lut, CubeK& irr,
                                        CubeK& pre, ShadeK& k)
{
    MIFX_REQUIRE(ibl != nullptr, "ibl must not be null");
    MIFX_REQUIRE(a.Workflow == MIFX_PBR_WORKFLOW_METALLIC_ROUGHNESS || a.Workflow == MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS, "unknown PBR workflow %d", a.Workflow);
    MIFX_CHECK(make_lutk(ibl->brdf_lut, lut));
    MIFX_CHECK(make_cubek(ibl->irradiance, "ibl.irradiance", irr));
    MIFX_CHECK(make_cubek(ibl->prefiltered, "ibl.prefiltered", pre));
    k.iblScale[0] = a.IBLScale[0]; k.iblScale[1] = a.IBLScale[1]; k.iblScale[2] = a.IBLScale[2];
    k.occlusionStrength = a.OcclusionStrength; k.emissionScale = a.EmissionScale; k.prefilteredCubeLastMip = a.PrefilteredCubeLastMip;
    k.lightCount = a.LightCount;
    k.workflow   = a.Workflow;
    for (int i = 0; i < a.LightCount; ++i) k.lights[i] = a.Lights[i];
    for (int i = 0; i < 4; ++i) k.background[i] = background ? background[i] : 0.0f;
    // working copies with face aprons (8.3 MB for a 256^2 prefiltered cube: ~10 us per call, repaid many times over in the shade kernel)
    const size_t irrBytes = apron_bytes(irr.size, 1), preBytes = apron_bytes(pre.size, pre.mips);
    DeviceScratch& iblApron = cache.scratch;
    const void* const oldData = iblApron.data;
    MIFX_CHECK(iblApron.reserve(irrBytes + preBytes));
    CubeK irrA, preA;
    ApronPlan irrPlan, prePlan;
    const int irrBlocks = plan_cube_apron(irr, 1, static_cast<unsigned char*>(iblApron.data), irrA, irrPlan); // sampled at lod 0 only
    const int preBlocks = plan_cube_apron(pre, pre.mips, static_cast<unsigned char*>(iblApron.data) + irrBytes, preA, prePlan);
    // The copy is made per call (the maps are the caller's memory: they may have been re-rendered) unless the caller declared them static
    // (mifx_postfx_set_static_ibl): then it is made once per set of maps (addresses + sizes).
    const void* key[26] = {};
    key[0] = irr.mip[0]; key[1] = reinterpret_cast<const void*>(uintptr_t(irr.size)); key[2] = reinterpret_cast<const void*>(uintptr_t(pre.size));
    key[3] = reinterpret_cast<const void*>(uintptr_t(pre.mips));
    for (int i = 0; i < pre.mips && i < 12; ++i) key[4 + i] = pre.mip[i];
    const bool reuse = cache.keep && cache.valid && oldData == iblApron.data && std::memcmp(key, cache.key, sizeof(key)) == 0;
    if (!reuse)
    {
        hipLaunchKernelGGL(cube_apron_kernel, dim3(irrBlocks + preBlocks, 1, 1), dim3(256, 1, 1), 0, s, irr, irrPlan, irrBlocks, pre, prePlan);
        MIFX_HIP_CHECK(hipGetLastError());
        std::memcpy(cache.key, key, sizeof(key));
        cache.valid = true;
    }
    irr = irrA;
    pre = preA;
    return MIFX_OK;
}

static mifx_status make_shadowk(const mifx_pbr_shadows& shadows, ShadowK& sh)
{
    const mifx_shadow_map_array* m = shadows.shadow_map;
    MIFX_REQUIRE(m != nullptr && m->data != nullptr && m->width > 0 && m->height > 0 && m->slices > 0 && m->pitch_bytes >= m->width * 4u && m->pitch_bytes % 4u == 0 &&
                     m->slice_pitch_bytes >= uint64_t(m->pitch_bytes) * m->height,
                 "shadows: bad shadow-map array");
    MIFX_REQUIRE(shadows.shadow_map_count <= MIFX_PBR_MAX_SHADOW_MAPS && (shadows.shadow_map_count == 0 || shadows.shadow_maps != nullptr), "shadows: %u shadow maps (at most %d)",
                 shadows.shadow_map_count, MIFX_PBR_MAX_SHADOW_MAPS);
    MIFX_REQUIRE(shadows.pcf_filter_size == 2 || shadows.pcf_filter_size == 3 || shadows.pcf_filter_size == 5 || shadows.pcf_filter_size == 7,
                 "shadows: PCF filter size %u (2, 3, 5 or 7: PCF.fxh)", shadows.pcf_filter_size);
    sh.data = static_cast<const unsigned char*>(m->data);
    sh.w = int(m->width); sh.h = int(m->height); sh.slices = int(m->slices); sh.pitch = int(m->pitch_bytes);
    sh.slicePitch = m->slice_pitch_bytes;
    sh.pcf = int(shadows.pcf_filter_size);
    for (uint32_t i = 0; i < shadows.shadow_map_count; ++i) sh.info[i] = shadows.shadow_maps[i];
    return MIFX_OK;
}

mifx_status launch_pbr_shade(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl,
                             const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_spec, int row_begin, int row_end, bool reversedDepth,
                             const mifx_pbr_shadows* shadows, const SsrMaskOut* ssrMask)
{
    MIFX_REQUIRE(ssrMask == nullptr || shadows == nullptr, "launch_pbr_shade: the SSR mask output is not combined with the shadowed permutation");
    const SsrMaskOut r2 = ssrMask ? *ssrMask : SsrMaskOut{};
    Img bc, nrm, mat, depth, emis{}, occ{}, outR, outS{};
    MIFX_CHECK(to_img(out_radiance, MIFX_FORMAT_F32X4, "out_radiance", outR));
    outR = rows_of(outR, row_begin, row_end);
    const uint32_t W = out_radiance->width, H = out_radiance->height;
    MIFX_CHECK(to_img_wh(g->base_color, MIFX_FORMAT_F32X4, W, H, "gbuffer.base_color", bc));
    MIFX_CHECK(to_img_wh(g->normal, MIFX_FORMAT_F32X4, W, H, "gbuffer.normal", nrm));
    MIFX_CHECK(to_img_wh(g->material, MIFX_FORMAT_F32X4, W, H, "gbuffer.material", mat));
    MIFX_CHECK(to_img_wh(g->depth, MIFX_FORMAT_F32, W, H, "gbuffer.depth", depth));
    if (g->emissive) MIFX_CHECK(to_img_wh(g->emissive, MIFX_FORMAT_F32X4, W, H, "gbuffer.emissive", emis));
    if (g->occlusion) MIFX_CHECK(to_img_wh(g->occlusion, MIFX_FORMAT_F32, W, H, "gbuffer.occlusion", occ));
    if (out_spec) MIFX_CHECK(to_img_wh(out_spec, MIFX_FORMAT_F32X4, W, H, "out_specular_ibl", outS));
    MIFX_REQUIRE(a.LightCount >= 0 && a.LightCount <= MIFX_PBR_MAX_LIGHTS, "LightCount %d out of range", a.LightCount);
    for (int i = 0; i < a.LightCount; ++i)
    {
        MIFX_REQUIRE(a.Lights[i].Type >= 1 && a.Lights[i].Type <= 3, "light %d: unknown type %d", i, a.Lights[i].Type);
        if (a.Lights[i].ShadowMapIndex >= 0)
            MIFX_REQUIRE(shadows != nullptr && uint32_t(a.Lights[i].ShadowMapIndex) < shadows->shadow_map_count,
                         "light %d: ShadowMapIndex %d needs mifx_pbr_shade_execute_with_shadows and an entry in shadow_maps", i, a.Lights[i].ShadowMapIndex);
    }
    ShadowK sh{};
    if (shadows != nullptr) MIFX_CHECK(make_shadowk(*shadows, sh));
    LutK lut;
    CubeK irr, pre;
    ShadeK k{};
    MIFX_CHECK(make_shade_constants(s, iblApron, a, ibl, background, lut, irr, pre, k));
    const CamK cam = make_camk(camera, reversedDepth);
    const dim3 block(64, 4, 1), grid = grid2d(outR, block);
#define MIFX_SHADE(E, A, S)                                                                                                                                  \
    if (shadows) hipLaunchKernelGGL((pbr_shade_shadowed_kernel<E, A, S>), grid, block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, outR, outS, cam, k, sh); \
    else hipLaunchKernelGGL((pbr_shade_kernel<E, A, S>), grid, block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, outR, outS, cam, k, r2)
    const int sel = (g->emissive ? 4 : 0) | (g->occlusion ? 2 : 0) | (out_spec ? 1 : 0);
    switch (sel)
    {
        case 0: MIFX_SHADE(false, false, false); break;
        case 1: MIFX_SHADE(false, false, true); break;
        case 2: MIFX_SHADE(false, true, false); break;
        case 3: MIFX_SHADE(false, true, true); break;
        case 4: MIFX_SHADE(true, false, false); break;
        case 5: MIFX_SHADE(true, false, true); break;
        case 6: MIFX_SHADE(true, true, false); break;
        default: MIFX_SHADE(true, true, true); break;
    }
#undef MIFX_SHADE
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

// one channel of a look-up table (the two sheen tables)
static mifx_status make_lutk_r(const mifx_image2d* im, const char* what, LutK& k)
{
    MIFX_REQUIRE(im != nullptr && im->data != nullptr && im->width > 0 && im->height > 0 &&
                     (im->format == MIFX_FORMAT_F32 || im->format == MIFX_FORMAT_F32X2 || im->format == MIFX_FORMAT_F32X4),
                 "%s: F32, F32X2 or F32X4 image required", what);
    k.data   = static_cast<const float*>(im->data);
    k.size_w = int(im->width);
    k.size_h = int(im->height);
    k.comps  = im->format == MIFX_FORMAT_F32 ? 1 : im->format == MIFX_FORMAT_F32X2 ? 2 : 4;
    MIFX_REQUIRE(im->pitch_bytes % 4u == 0 && im->pitch_bytes >= im->width * 4u * uint32_t(k.comps), "%s: bad pitch", what);
    k.pitch_f = int(im->pitch_bytes / 4u);
    return MIFX_OK;
}
mifx_status launch_pbr_shade_layers(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_pbr_layers& layers, const mifx_camera_attribs& camera,
                                    const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_spec,
                                    int row_begin, int row_end, bool reversedDepth, const mifx_pbr_shadows* shadows, const LayeredHitFetch* hit)
{
    const uint32_t known = MIFX_PBR_LAYER_CLEAR_COAT | MIFX_PBR_LAYER_SHEEN | MIFX_PBR_LAYER_ANISOTROPY | MIFX_PBR_LAYER_IRIDESCENCE | MIFX_PBR_LAYER_TRANSMISSION;
    MIFX_REQUIRE((layers.flags & ~known) == 0u, "layers: unknown flag bits 0x%x", layers.flags & ~known);
    Img bc, nrm, mat, depth, emis{}, occ{}, outR, outS{};
    MIFX_CHECK(to_img(out_radiance, MIFX_FORMAT_F32X4, "out_radiance", outR)); // (hit fetch: the plane the band's rows were shaded into)
    if (hit == nullptr) outR = rows_of(outR, row_begin, row_end);
    const uint32_t W = out_radiance->width, H = out_radiance->height;
    if (hit != nullptr) MIFX_REQUIRE(W <= 65536u && H <= 65536u, "launch_pbr_shade_layers: frame %ux%u exceeds the 16-bit hit coordinates", W, H);
    MIFX_CHECK(to_img_wh(g->base_color, MIFX_FORMAT_F32X4, W, H, "gbuffer.base_color", bc));
    MIFX_CHECK(to_img_wh(g->normal, MIFX_FORMAT_F32X4, W, H, "gbuffer.normal", nrm));
    MIFX_CHECK(to_img_wh(g->material, MIFX_FORMAT_F32X4, W, H, "gbuffer.material", mat));
    MIFX_CHECK(to_img_wh(g->depth, MIFX_FORMAT_F32, W, H, "gbuffer.depth", depth));
    if (g->emissive) MIFX_CHECK(to_img_wh(g->emissive, MIFX_FORMAT_F32X4, W, H, "gbuffer.emissive", emis));
    if (g->occlusion) MIFX_CHECK(to_img_wh(g->occlusion, MIFX_FORMAT_F32, W, H, "gbuffer.occlusion", occ));
    if (out_spec) MIFX_CHECK(to_img_wh(out_spec, MIFX_FORMAT_F32X4, W, H, "out_specular_ibl", outS));
    MIFX_REQUIRE(a.LightCount >= 0 && a.LightCount <= MIFX_PBR_MAX_LIGHTS, "LightCount %d out of range", a.LightCount);
    for (int i = 0; i < a.LightCount; ++i)
    {
        MIFX_REQUIRE(a.Lights[i].Type >= 1 && a.Lights[i].Type <= 3, "light %d: unknown type %d", i, a.Lights[i].Type);
        if (a.Lights[i].ShadowMapIndex >= 0)
            MIFX_REQUIRE(shadows != nullptr && uint32_t(a.Lights[i].ShadowMapIndex) < shadows->shadow_map_count, "light %d: ShadowMapIndex %d needs `shadows` and an entry in shadow_maps", i,
                         a.Lights[i].ShadowMapIndex);
    }
    ShadowK sh{};
    if (shadows != nullptr) MIFX_CHECK(make_shadowk(*shadows, sh));
    LayersK ly{};
    ly.flags = layers.flags;
    if (layers.flags & MIFX_PBR_LAYER_CLEAR_COAT)
    {
        MIFX_CHECK(to_img_wh(layers.clearcoat, MIFX_FORMAT_F32X4, W, H, "layers.clearcoat", ly.clearcoat));
        if (layers.clearcoat_normal) MIFX_CHECK(to_img_wh(layers.clearcoat_normal, MIFX_FORMAT_F32X4, W, H, "layers.clearcoat_normal", ly.clearcoatNormal));
        ly.hasClearcoatNormal = layers.clearcoat_normal != nullptr;
    }
    if (layers.flags & MIFX_PBR_LAYER_SHEEN)
    {
        MIFX_CHECK(to_img_wh(layers.sheen, MIFX_FORMAT_F32X4, W, H, "layers.sheen", ly.sheen));
        MIFX_CHECK(make_lutk_r(layers.sheen_albedo_scaling_lut, "layers.sheen_albedo_scaling_lut", ly.albedoScaling));
        MIFX_CHECK(make_lutk_r(layers.preintegrated_charlie, "layers.preintegrated_charlie", ly.charlie));
    }
    if (layers.flags & MIFX_PBR_LAYER_ANISOTROPY)
    {
        MIFX_CHECK(to_img_wh(layers.anisotropy, MIFX_FORMAT_F32X4, W, H, "layers.anisotropy", ly.anisotropy));
        if (layers.tangent) MIFX_CHECK(to_img_wh(layers.tangent, MIFX_FORMAT_F32X4, W, H, "layers.tangent", ly.tangent));
        ly.hasTangent  = layers.tangent != nullptr;
        ly.rotationCos = std::cos(layers.anisotropy_rotation); // float overloads: the shader's cos / sin of a per-material constant (RenderPBR.psh:263)
        ly.rotationSin = std::sin(layers.anisotropy_rotation);
    }
    if (layers.flags & MIFX_PBR_LAYER_IRIDESCENCE)
    {
        MIFX_CHECK(to_img_wh(layers.iridescence, MIFX_FORMAT_F32X4, W, H, "layers.iridescence", ly.iridescence));
        MIFX_REQUIRE(layers.iridescence_ior > 0.0f, "layers.iridescence_ior %g", double(layers.iridescence_ior));
        ly.iridescenceIor = layers.iridescence_ior;
    }
    if (layers.flags & MIFX_PBR_LAYER_TRANSMISSION) MIFX_CHECK(to_img_wh(layers.transmission, MIFX_FORMAT_F32, W, H, "layers.transmission", ly.transmission));
    LutK lut;
    CubeK irr, pre;
    ShadeK k{};
    MIFX_CHECK(make_shade_constants(s, iblApron, a, ibl, background, lut, irr, pre, k));
    const CamK cam = make_camk(camera, reversedDepth);
    const dim3 block(64, 4, 1);
    if (hit != nullptr)
    {
        const HitOut out{hit->rays, hit->coords, outR, hit->shadedBegin, hit->shadedEnd};
        hipLaunchKernelGGL(shadows ? pbr_hit_fetch_layers_kernel<true> : pbr_hit_fetch_layers_kernel<false>, grid2d(hit->rays, block), block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr,
                           pre, out, cam, k, ly, g->emissive ? 1 : 0, g->occlusion ? 1 : 0, sh);
    }
    else
    {
        // each single layer and all five have an instance of their own; any other set takes the run-time one (MIFX_LAYERS_GENERIC=1 forces it: A/B, tests)
        static const bool generic = [] { const char* e = std::getenv("MIFX_LAYERS_GENERIC"); return e && std::atoi(e) != 0; }();
#define MIFX_LAYERS_INSTANCE(SET) (shadows ? pbr_shade_layers_kernel<SET, true> : pbr_shade_layers_kernel<SET, false>)
        auto* kernel = MIFX_LAYERS_INSTANCE(kLayersRuntime);
        if (!generic) switch (layers.flags)
        {
            case MIFX_PBR_LAYER_CLEAR_COAT: kernel = MIFX_LAYERS_INSTANCE(MIFX_PBR_LAYER_CLEAR_COAT); break;
            case MIFX_PBR_LAYER_SHEEN: kernel = MIFX_LAYERS_INSTANCE(MIFX_PBR_LAYER_SHEEN); break;
            case MIFX_PBR_LAYER_ANISOTROPY: kernel = MIFX_LAYERS_INSTANCE(MIFX_PBR_LAYER_ANISOTROPY); break;
            case MIFX_PBR_LAYER_IRIDESCENCE: kernel = MIFX_LAYERS_INSTANCE(MIFX_PBR_LAYER_IRIDESCENCE); break;
            case MIFX_PBR_LAYER_TRANSMISSION: kernel = MIFX_LAYERS_INSTANCE(MIFX_PBR_LAYER_TRANSMISSION); break;
            case 31u: kernel = MIFX_LAYERS_INSTANCE(31u); break;
            default: break;
        }
#undef MIFX_LAYERS_INSTANCE
        hipLaunchKernelGGL(kernel, grid2d(outR, block), block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, outR, outS, cam, k, ly, g->emissive ? 1 : 0, g->occlusion ? 1 : 0,
                           out_spec ? 1 : 0, sh);
    }
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

mifx_status launch_pbr_hit_fetch(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl,
                                 const float background[4], Img rays, Img hitCoords, const mifx_image2d* radiance, int shadedBegin, int shadedEnd, bool reversedDepth)
{
    Img bc, nrm, mat, depth, emis{}, occ{}, rad;
    MIFX_CHECK(to_img(radiance, MIFX_FORMAT_F32X4, "radiance", rad));
    const uint32_t W = radiance->width, H = radiance->height;
    MIFX_REQUIRE(W <= 65536u && H <= 65536u, "launch_pbr_hit_fetch: frame %ux%u exceeds the 16-bit hit coordinates", W, H);
    MIFX_CHECK(to_img_wh(g->base_color, MIFX_FORMAT_F32X4, W, H, "gbuffer.base_color", bc));
    MIFX_CHECK(to_img_wh(g->normal, MIFX_FORMAT_F32X4, W, H, "gbuffer.normal", nrm));
    MIFX_CHECK(to_img_wh(g->material, MIFX_FORMAT_F32X4, W, H, "gbuffer.material", mat));
    MIFX_CHECK(to_img_wh(g->depth, MIFX_FORMAT_F32, W, H, "gbuffer.depth", depth));
    if (g->emissive) MIFX_CHECK(to_img_wh(g->emissive, MIFX_FORMAT_F32X4, W, H, "gbuffer.emissive", emis));
    if (g->occlusion) MIFX_CHECK(to_img_wh(g->occlusion, MIFX_FORMAT_F32, W, H, "gbuffer.occlusion", occ));
    LutK lut;
    CubeK irr, pre;
    ShadeK k{};
    MIFX_CHECK(make_shade_constants(s, iblApron, a, ibl, background, lut, irr, pre, k));
    const CamK cam = make_camk(camera, reversedDepth);
    const HitOut out{rays, hitCoords, rad, shadedBegin, shadedEnd};
    const dim3 block(64, 4, 1), grid = grid2d(rays, block);
#define MIFX_FETCH(E, A) hipLaunchKernelGGL((pbr_hit_fetch_kernel<E, A>), grid, block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, out, cam, k)
    switch ((g->emissive ? 2 : 0) | (g->occlusion ? 1 : 0))
    {
        case 0: MIFX_FETCH(false, false); break;
        case 1: MIFX_FETCH(false, true); break;
        case 2: MIFX_FETCH(true, false); break;
        default: MIFX_FETCH(true, true); break;
    }
#undef MIFX_FETCH
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

// The shade on the reference's own G-buffer formats: same kernel body, the format conversion is the load / store (no fp32 copies of the planes in HBM)
mifx_status launch_pbr_shade_native(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer_native* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a,
                                    const mifx_ibl* ibl, const float background[4], const mifx_native_image* out_radiance, const mifx_native_image* out_spec, bool reversedDepth)
{
    NativeImg bc, nrm, mat, depth, emis{}, occ{}, outR, outS{};
    MIFX_CHECK(to_native(out_radiance, "out_radiance", outR));
    MIFX_CHECK(to_native(g->base_color, "gbuffer.base_color", bc));
    MIFX_CHECK(to_native(g->normal, "gbuffer.normal", nrm));
    MIFX_CHECK(to_native(g->material, "gbuffer.material", mat));
    MIFX_CHECK(to_native(g->depth, "gbuffer.depth", depth));
    if (g->emissive) MIFX_CHECK(to_native(g->emissive, "gbuffer.emissive", emis));
    if (g->occlusion) MIFX_CHECK(to_native(g->occlusion, "gbuffer.occlusion", occ));
    if (out_spec) MIFX_CHECK(to_native(out_spec, "out_specular_ibl", outS));
    MIFX_REQUIRE(depth.fmt == MIFX_NATIVE_FORMAT_R32_FLOAT, "gbuffer.depth: the depth buffer is R32_FLOAT (D32_FLOAT read as a colour plane), got format %u", depth.fmt);
    for (const NativeImg* i : {&bc, &nrm, &mat, &depth, g->emissive ? &emis : &outR, g->occlusion ? &occ : &outR, out_spec ? &outS : &outR})
        MIFX_REQUIRE(i->w == outR.w && i->h == outR.h, "native shade: plane of %dx%d, output %dx%d", i->w, i->h, outR.w, outR.h);
    MIFX_REQUIRE(a.LightCount >= 0 && a.LightCount <= MIFX_PBR_MAX_LIGHTS, "LightCount %d out of range", a.LightCount);
    for (int i = 0; i < a.LightCount; ++i)
    {
        MIFX_REQUIRE(a.Lights[i].Type >= 1 && a.Lights[i].Type <= 3, "light %d: unknown type %d", i, a.Lights[i].Type);
        MIFX_REQUIRE(a.Lights[i].ShadowMapIndex < 0, "light %d: shadow maps go through mifx_pbr_shade_execute_with_shadows", i);
    }
    LutK lut;
    CubeK irr, pre;
    ShadeK k{};
    MIFX_CHECK(make_shade_constants(s, iblApron, a, ibl, background, lut, irr, pre, k));
    const CamK cam = make_camk(camera, reversedDepth);
    const dim3 block(64, 4, 1), grid = grid2d(outR.w, outR.h, block);
    // Hydrogent's own formats (no extra emissive / occlusion planes, whose formats are the caller's) can take the per-format instance (opt-in, see the kernel)
    const bool hydrogent = !g->emissive && !g->occlusion && bc.fmt == MIFX_NATIVE_FORMAT_RGBA8_UNORM && nrm.fmt == MIFX_NATIVE_FORMAT_RGBA16_FLOAT &&
                           mat.fmt == MIFX_NATIVE_FORMAT_RG8_UNORM && outR.fmt == MIFX_NATIVE_FORMAT_RGBA16_FLOAT && (!out_spec || outS.fmt == MIFX_NATIVE_FORMAT_RGBA16_FLOAT) &&
                           std::getenv("MIFX_NATIVE_SHADE_PER_FORMAT") != nullptr;
#define MIFX_SHADE(E, A, S) hipLaunchKernelGGL((pbr_shade_native_kernel<E, A, S, false>), grid, block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, outR, outS, cam, k)
    if (hydrogent)
    {
        if (out_spec) hipLaunchKernelGGL((pbr_shade_native_kernel<false, false, true, true>), grid, block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, outR, outS, cam, k);
        else hipLaunchKernelGGL((pbr_shade_native_kernel<false, false, false, true>), grid, block, 0, s, bc, nrm, mat, depth, emis, occ, lut, irr, pre, outR, outS, cam, k);
    }
    else
    switch ((g->emissive ? 4 : 0) | (g->occlusion ? 2 : 0) | (out_spec ? 1 : 0))
    {
        case 0: MIFX_SHADE(false, false, false); break;
        case 1: MIFX_SHADE(false, false, true); break;
        case 2: MIFX_SHADE(false, true, false); break;
        case 3: MIFX_SHADE(false, true, true); break;
        case 4: MIFX_SHADE(true, false, false); break;
        case 5: MIFX_SHADE(true, false, true); break;
        case 6: MIFX_SHADE(true, true, false); break;
        default: MIFX_SHADE(true, true, true); break;
    }
#undef MIFX_SHADE
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

// The Material target of the USD G-buffer for a specular-glossiness surface (USD_Renderer.cpp:98: (Srf.PerceptualRoughness, BaseLayer.Metallic))
__global__ __launch_bounds__(256) void specgloss_material_kernel(Img baseColor, Img physicalDesc, Img out)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    float metallic = 0.0f;
    const SurfaceReflectance srf = surface_reflectance_workflow_sg(xyz(ld<v4>(baseColor, x, y)), ld<v4>(physicalDesc, x, y), metallic);
    st<v4>(out, x, y, v4{srf.perceptualRoughness, metallic, 0.0f, 0.0f});
}
mifx_status launch_specgloss_material(hipStream_t s, Img baseColor, Img physicalDesc, Img out)
{
    const dim3 block(64, 4, 1);
    hipLaunchKernelGGL(specgloss_material_kernel, grid2d(out, block), block, 0, s, baseColor, physicalDesc, out);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

} // namespace mifx
