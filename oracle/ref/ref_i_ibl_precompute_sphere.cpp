// TEST INFRASTRUCTURE ONLY (oracle/_ref).  I2 / I3 with an equirectangular ("sphere") environment map (Macros: ENV_MAP_TYPE = ENV_MAP_TYPE_SPHERE, PBR_Renderer.cpp:788-800):
// the environment is a Texture2D sampled at TransformDirectionToSphereMapUV(dir) (ShaderUtilities.fxh:98-102) with the linear-clamp sampler (:851, 906), and the
// solid angle of a pixel follows ComputeSphereMapPixelSolidAngle (PBR_PrecomputeCommon.fxh:43-48).  This is the reference's equirect -> cube path.
#include "ref_common.h"
#define ENV_MAP_TYPE_CUBE 0
#define ENV_MAP_TYPE_SPHERE 1
#define ENV_MAP_TYPE ENV_MAP_TYPE_SPHERE
#define OPTIMIZE_SAMPLES 1
namespace hlsl {
namespace i2s {
#include "ShaderDefinitions.fxh"
#include "PrefilterEnvMap.psh"
}
#undef _PBR_PRECOMPUTE_COMMON_FXH_
#undef _PBR_COMMON_FXH_
#undef _SHADER_UTILITIES_FXH_
namespace i3s {
#include "ComputeIrradianceMap.psh"
}
}
using namespace hlsl;

template <class F> static void for_each_cube_texel(const ref_img& o, F&& f)
{
    const int n = o.w;
#pragma omp parallel for schedule(dynamic, 1)
    for (int row = 0; row < 6 * n; ++row)
        for (int x = 0; x < n; ++x)
        {
            int face = row / n, y = row % n;
            float3 d = normalize(hl_cube_dir(face, (float(x) + 0.5f) / float(n), (float(y) + 0.5f) / float(n)));
            f(x, row, d);
        }
}

// in[0]: sphere map (2D mip chain, c=4); out[0]: one mip of the prefiltered cube (w x 6w, c=4); fval[0]: roughness; ival[0]: samples
extern "C" int ref_ibl_prefilter_env_map_sphere(const ref_args* a)
{
    ref_bind(i2s::g_EnvironmentMap.s, a, 0);
    i2s::g_EnvironmentMap_sampler = Sam_LinearClamp;
    i2s::g_Roughness    = a->fval[0];
    i2s::g_EnvMapWidth  = float(a->in[0][0].w);
    i2s::g_EnvMapHeight = float(a->in[0][0].h);
    i2s::g_EnvMipCount  = float(a->in_mips[0]);
    i2s::g_NumSamples   = unsigned(a->ival[0]);
    const ref_img& o = a->out[0];
    for_each_cube_texel(o, [&](int x, int row, const float3& d) { ref_store(o, x, row, float4(i2s::PrefilterEnvMap(i2s::g_Roughness, d), 0.0f)); });
    return 0;
}
// in[0]: sphere map (2D mip chain); out[0]: irradiance cube (w x 6w, c=4); ival[0]: samples
extern "C" int ref_ibl_irradiance_map_sphere(const ref_args* a)
{
    ref_bind(i3s::g_EnvironmentMap.s, a, 0);
    i3s::g_EnvironmentMap_sampler = Sam_LinearClamp;
    i3s::g_EnvMapWidth  = float(a->in[0][0].w);
    i3s::g_EnvMapHeight = float(a->in[0][0].h);
    i3s::g_EnvMipCount  = float(a->in_mips[0]);
    i3s::g_NumSamples   = unsigned(a->ival[0]);
    const ref_img& o = a->out[0];
    for_each_cube_texel(o, [&](int x, int row, const float3& d) { ref_store(o, x, row, float4(i3s::IrradianceMap(d), 1.0f)); });
    return 0;
}
