// TEST INFRASTRUCTURE ONLY (oracle/_ref).  P1-P9: the lighting half of the PBR pixel shader, evaluated per pixel from a G-buffer.
// Every lighting function is the reference's own code (Shaders/PBR/public/PBR_Shading.fxh, Shaders/Common/public/PBR_Common.fxh);
// this wrapper only restates the call sequence of Shaders/PBR/private/RenderPBR.psh:
//   GetSurfaceShadingInfo (:299-359)  -> ReadBaseLayerProperties (:138-184, metallic-roughness branch, factors = 1)
//   for lights: ApplyPunctualLight (:479-499) ; ApplyIBL (:501-512) ; ResolveLighting (:514)
// with the material-fetch half replaced by the G-buffer (contract: PBR/src/USD_Renderer.cpp:83-162: Normal = shading normal,
// Material = (roughness, metallic), IBL target = GetBaseLayerSpecularIBL) and the world position rebuilt from depth with
// InvProjectPosition (PostFX_Common.fxh:99-105), the in-repo pattern of SSR_ComputeSpatialReconstruction.fx:108-111.
#include "ref_common.h"
#define PBR_MAX_LIGHTS 16
#define USE_IBL 1
#ifndef ENABLE_SHADOWS // ref_p_pbr_shade_shadows*.cpp build the shadowed permutations (PCF_FILTER_SIZE 3 and 5) of this file
#define ENABLE_SHADOWS 0
#endif
namespace hlsl { namespace pbr {
#include "ShaderDefinitions.fxh"
#include "BasicStructures.fxh"
#include "PostFX_Common.fxh"
#include "PBR_Structures.fxh"
#include "RenderPBR_Structures.fxh"
#include "PBR_Shading.fxh"

Texture2D_<float4> g_BaseColor, g_Normal, g_Material, g_Emissive;
Texture2D_<float>  g_Depth, g_Occlusion;
Texture2D_<float4> g_PreintegratedGGX;
TextureCube        g_IrradianceMap, g_PrefilteredEnvMap;
#if ENABLE_SHADOWS
Texture2DArray_<float> g_ShadowMap; // RenderPBR.psh:70-73
SamplerComparisonState g_ShadowMap_sampler;
#endif
}}
using namespace hlsl;

struct ShadeAttribs // == mifx_pbr_shade_attribs (include/mifx.h)
{
    float IBLScale[4];
    float OcclusionStrength, EmissionScale, PrefilteredCubeLastMip;
    int   LightCount;
    pbr::PBRLightAttribs Lights[16];
    int   Workflow, Padding[3];
};

// in: 0 base colour (c=4), 1 normal (c=4), 2 material (c=4: roughness, metallic), 3 depth, 4 emissive (c=4) or none, 5 occlusion or none,
//     6 BRDF LUT (c=2 or 4), 7 irradiance cube (faces stacked: w x 6w, c=4), 8 prefiltered cube (mips); cam0; attribs: ShadeAttribs
//     fval[0..3]: background colour.  out: 0 radiance (c=4), 1 specular IBL (c=4)
//     shadowed permutations: in[9] = the slices of the shadow-map array (one "mip" per slice, all the same size), in[10] = n x 24 floats, the PBRShadowMapInfo
//     array of the frame attribs (RenderPBR_Structures.fxh:22); a light with ShadowMapIndex >= 0 is attenuated by FilterShadowMapFixedPCF (PBR_Shading.fxh:644-660)
extern "C" int ref_pbr_shade(const ref_args* a)
{
    ref_bind(pbr::g_BaseColor.s, a, 0);
    ref_bind(pbr::g_Normal.s, a, 1);
    ref_bind(pbr::g_Material.s, a, 2);
    ref_bind(pbr::g_Depth.s, a, 3);
    const bool has_emissive = a->in_mips[4] > 0, has_ao = a->in_mips[5] > 0;
    if (has_emissive) ref_bind(pbr::g_Emissive.s, a, 4);
    if (has_ao) ref_bind(pbr::g_Occlusion.s, a, 5);
    ref_bind(pbr::g_PreintegratedGGX.s, a, 6);
    ref_bind_cube(pbr::g_IrradianceMap.s, a, 7);
    ref_bind_cube(pbr::g_PrefilteredEnvMap.s, a, 8);
    pbr::CameraAttribs cam;
    std::memcpy(&cam, a->cam0, sizeof(cam));
    ShadeAttribs sa;
    std::memcpy(&sa, a->attribs, sizeof(sa));
    const float4 background(a->fval[0], a->fval[1], a->fval[2], a->fval[3]);
#if ENABLE_SHADOWS
    {
        const float* slices[32];
        const int n = a->in_mips[9];
        for (int i = 0; i < n; ++i) slices[i] = a->in[9][i].data;
        ref_bind_array(pbr::g_ShadowMap, slices, n, a->in[9][0].w, a->in[9][0].h);
        pbr::g_ShadowMap_sampler = Sam_ComparisonLinearClamp;
    }
    const pbr::PBRShadowMapInfo* shadowInfos = reinterpret_cast<const pbr::PBRShadowMapInfo*>(a->in[10][0].data);
    static_assert(sizeof(pbr::PBRShadowMapInfo) == 96, "PBRShadowMapInfo layout");
#endif
    const SamplerState linear = Sam_LinearClamp;
    const ref_img &o0 = a->out[0], &o1 = a->out[1];
    const int W = o0.w, H = o0.h;
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
        {
            int3  pc(x, y, 0);
            float depth = pbr::g_Depth.Load(pc);
            if (a->ival[7] ? depth < 1e-6f : depth >= 1.0f - 1e-6f) // background (ival[7]: reversed depth)
            {
                ref_store(o0, x, y, background);
                if (o1.data) ref_store(o1, x, y, float4(0.f, 0.f, 0.f, 0.f));
                continue;
            }
            float4 BaseColor = pbr::g_BaseColor.Load(pc);
            float4 Material  = pbr::g_Material.Load(pc);
            float2 uv((float(x) + 0.5f) * cam.f4ViewportSize.z, (float(y) + 0.5f) * cam.f4ViewportSize.w);

            pbr::SurfaceShadingInfo Shading;
            Shading.Pos  = pbr::InvProjectPosition(float3(uv, depth), cam.mViewProjInv);
            Shading.View = normalize(cam.f4Position.xyz - Shading.Pos);
            // ReadBaseLayerProperties (RenderPBR.psh:151-173), factors = 1: metallic-roughness reads (roughness, metallic) of the USD Material target into .g / .b;
            // specular-glossiness takes the plane as the fetched PhysicalDesc and converts the specular colour with FastSRGBToLinear
            float4 PhysicalDesc(0.0f, Material.x, Material.y, 0.0f);
            if (sa.Workflow == PBR_WORKFLOW_SPECULAR_GLOSSINESS)
            {
                PhysicalDesc = Material;
                PhysicalDesc.rgb = pbr::FastSRGBToLinear(PhysicalDesc.rgb);
                PhysicalDesc.r *= 1.0f; PhysicalDesc.g *= 1.0f; PhysicalDesc.b *= 1.0f;
                PhysicalDesc.a *= 1.0f;
            }
            else
            {
                PhysicalDesc.g = saturate(PhysicalDesc.g * 1.0f);
                PhysicalDesc.b = saturate(PhysicalDesc.b * 1.0f);
            }
            Shading.BaseLayer.Metallic = 0.0f;
            Shading.BaseLayer.Srf      = pbr::GetSurfaceReflectance(sa.Workflow, BaseColor, PhysicalDesc, Shading.BaseLayer.Metallic);
            Shading.BaseLayer.Normal   = pbr::g_Normal.Load(pc).xyz;
            Shading.BaseLayer.NdotV    = pbr::dot_sat(Shading.BaseLayer.Normal, Shading.View);
            Shading.Occlusion = has_ao ? pbr::g_Occlusion.Load(pc) : 1.0f;
            Shading.Emissive  = has_emissive ? float3(pbr::g_Emissive.Load(pc).xyz) : float3(0.f, 0.f, 0.f);
            Shading.IBLScale  = float3(sa.IBLScale[0], sa.IBLScale[1], sa.IBLScale[2]);
            Shading.Occlusion = lerp(1.0f, Shading.Occlusion, sa.OcclusionStrength);
            Shading.Emissive *= sa.EmissionScale;

            pbr::SurfaceLightingInfo SrfLighting = pbr::GetDefaultSurfaceLightingInfo();
            int LightCount = min(sa.LightCount, PBR_MAX_LIGHTS);
#if ENABLE_SHADOWS
            for (int i = 0; i < LightCount; ++i) // RenderPBR.psh:488-497
                pbr::ApplyPunctualLight(Shading, sa.Lights[i], pbr::g_ShadowMap, pbr::g_ShadowMap_sampler, shadowInfos[max(sa.Lights[i].ShadowMapIndex, 0)], SrfLighting);
#else
            for (int i = 0; i < LightCount; ++i) pbr::ApplyPunctualLight(Shading, sa.Lights[i], SrfLighting);
#endif
            pbr::ApplyIBL(Shading, sa.PrefilteredCubeLastMip, pbr::g_PreintegratedGGX, linear, pbr::g_IrradianceMap, linear, pbr::g_PrefilteredEnvMap, linear, SrfLighting);

            float3 color = pbr::ResolveLighting(Shading, SrfLighting);
            ref_store(o0, x, y, float4(color, BaseColor.a));
            if (o1.data) ref_store(o1, x, y, float4(pbr::GetBaseLayerSpecularIBL(Shading, SrfLighting), 1.0f));
        }
    return 0;
}
#if !ENABLE_SHADOWS
// The Material target of a specular-glossiness surface: USD_Renderer.cpp:98 writes (Srf.PerceptualRoughness, BaseLayer.Metallic) of the same ReadBaseLayerProperties result.
// in: 0 base colour (c=4), 1 PhysicalDesc (c=4); out: 0 (roughness, metallic, 0, 0)
extern "C" int ref_specgloss_material(const ref_args* a)
{
    ref_bind(pbr::g_BaseColor.s, a, 0);
    ref_bind(pbr::g_Material.s, a, 1);
    const ref_img& o = a->out[0];
    for (int y = 0; y < o.h; ++y)
        for (int x = 0; x < o.w; ++x)
        {
            int3   pc(x, y, 0);
            float4 PhysicalDesc = pbr::g_Material.Load(pc);
            PhysicalDesc.rgb = pbr::FastSRGBToLinear(PhysicalDesc.rgb);
            float Metallic = 0.0f;
            pbr::SurfaceReflectanceInfo Srf = pbr::GetSurfaceReflectance(PBR_WORKFLOW_SPECULAR_GLOSSINESS, pbr::g_BaseColor.Load(pc), PhysicalDesc, Metallic);
            ref_store(o, x, y, float4(Srf.PerceptualRoughness, Metallic, 0.0f, 0.0f));
        }
    return 0;
}
#endif
extern "C" int ref_sizeof_pbr_light_attribs() { return int(sizeof(pbr::PBRLightAttribs)); }
