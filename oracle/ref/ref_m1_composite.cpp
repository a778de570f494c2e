// TEST INFRASTRUCTURE ONLY (oracle/_ref).  M1: the SSR/SSAO composite of Hydrogent/shaders/HnPostProcess.psh:145-185, restated line by line
// on top of the reference's own functions (GetSurfaceReflectanceMR PBR_Shading.fxh:429, GetIBLSamplingInfo :232, GetSpecularIBL_GGX :293).
// The rest of that shader (selection outline, grid, edge map, desaturation) is editor UI and out of scope; tone mapping is applied by the
// copy-frame pass when TAA is on (HnPostProcessTask.cpp:172) and is therefore not part of this wrapper.
#include "ref_common.h"
namespace hlsl { namespace m1 {
#include "ShaderDefinitions.fxh"
#include "BasicStructures.fxh"
#include "PBR_Structures.fxh"
#include "PBR_Shading.fxh"
Texture2D_<float4> g_ColorBuffer, g_SSR, g_SSAO, g_Normal, g_SpecularIBL, g_MaterialData, g_BaseColor, g_PreintegratedGGX;
}}
using namespace hlsl;

// in: 0 colour, 1 specular IBL, 2 SSR, 3 SSAO (c=1), 4 normal, 5 base colour, 6 material, 7 BRDF LUT; cam0; fval[0] SSRScale, fval[1] SSAOScale; out[0]
extern "C" int ref_composite(const ref_args* a)
{
    ref_bind(m1::g_ColorBuffer.s, a, 0);
    ref_bind(m1::g_SpecularIBL.s, a, 1);
    ref_bind(m1::g_SSR.s, a, 2);
    ref_bind(m1::g_SSAO.s, a, 3);
    ref_bind(m1::g_Normal.s, a, 4);
    ref_bind(m1::g_BaseColor.s, a, 5);
    ref_bind(m1::g_MaterialData.s, a, 6);
    ref_bind(m1::g_PreintegratedGGX.s, a, 7);
    m1::CameraAttribs cam;
    std::memcpy(&cam, a->cam0, sizeof(cam));
    const float attrSSRScale = a->fval[0], attrSSAOScale = a->fval[1];
    const SamplerState linear = Sam_LinearClamp;
    const ref_img& o = a->out[0];
    const int W = o.w, H = o.h;
#pragma omp parallel for
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
        {
            int3   Pos(x, y, 0);
            float2 f2NormalizedXY(2.0f * (float(x) + 0.5f) / float(W) - 1.0f, 1.0f - 2.0f * (float(y) + 0.5f) / float(H));
            float4 Color   = m1::g_ColorBuffer.Load(Pos);
            float  Opacity = Color.a;
            float  SSRScale = attrSSRScale * Opacity;
            if (SSRScale > 0.0f)
            {
                float4 SpecularIBL = m1::g_SpecularIBL.Load(Pos);
                float4 SSRRadiance = m1::g_SSR.Load(Pos);
                float3 Normal      = m1::g_Normal.Load(Pos).xyz;
                float4 BaseColor   = m1::g_BaseColor.Load(Pos);
                float4 Material    = m1::g_MaterialData.Load(Pos);
                float Roughness = saturate(Material.x);
                float Metallic  = saturate(Material.y);
                m1::SurfaceReflectanceInfo SrfInfo = m1::GetSurfaceReflectanceMR(BaseColor.rgb, Metallic, Roughness);
                float4 WorldPos = mul(float4(f2NormalizedXY, DepthToNormalizedDeviceZ(0.5f), 1.0f), cam.mViewProjInv);
                float3 ViewDir  = normalize(cam.f4Position.xyz - WorldPos.xyz / WorldPos.w);
                m1::IBLSamplingInfo IBLInfo = m1::GetIBLSamplingInfo(SrfInfo, m1::g_PreintegratedGGX, linear, Normal, ViewDir);
                float3 SSR = m1::GetSpecularIBL_GGX(SrfInfo, IBLInfo, SSRRadiance.rgb);
                Color.rgb += (SSR.rgb - SpecularIBL.rgb) * SSRRadiance.w * SSRScale;
            }
            float SSAOScale = attrSSAOScale * Opacity;
            if (SSAOScale > 0.0f)
            {
                float Occlusion = lerp(1.0f, m1::g_SSAO.Load(Pos).x, SSAOScale);
                Color.rgb *= Occlusion;
            }
            ref_store(o, x, y, Color);
        }
    return 0;
}
