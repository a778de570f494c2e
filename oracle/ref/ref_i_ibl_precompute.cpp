// TEST INFRASTRUCTURE ONLY (oracle/_ref).  I1-I3: IBL precompute compiled from the reference source:
//   I1 Shaders/PBR/private/PrecomputeBRDF.psh        IntegrateBRDF :8-39  (PBR_Renderer::PrecomputeBRDF, PBR_Renderer.cpp:548-622, 512 samples)
//   I2 Shaders/PBR/private/PrefilterEnvMap.psh       PrefilterEnvMap :40-98 (mip -> roughness = mip/(mips-1), PBR_Renderer.cpp:951)
//   I3 Shaders/PBR/private/ComputeIrradianceMap.psh  IrradianceMap :43-83
// The cube-face rasterisation (CubemapFace.vsh + the rotation matrices of PBR_Renderer.cpp:905-913) is replaced by the direction of each
// texel centre in the D3D face convention (hlsl_cube_dir), which is what those matrices produce.
#include "ref_common.h"
#define ENV_MAP_TYPE_CUBE 0
#define ENV_MAP_TYPE_SPHERE 1
#define ENV_MAP_TYPE ENV_MAP_TYPE_CUBE
#define OPTIMIZE_SAMPLES 1
namespace hlsl {
namespace i1 {
#include "ShaderDefinitions.fxh"
#include "PrecomputeBRDF.psh"
}
// shared-header include guards are per TU: the second namespace needs the common functions again
#undef _PBR_PRECOMPUTE_COMMON_FXH_
#undef _PBR_COMMON_FXH_
#undef _SHADER_UTILITIES_FXH_
namespace i2 {
#include "PrefilterEnvMap.psh"
}
#undef _PBR_PRECOMPUTE_COMMON_FXH_
#undef _PBR_COMMON_FXH_
#undef _SHADER_UTILITIES_FXH_
namespace i3 {
#include "ComputeIrradianceMap.psh"
}
}
using namespace hlsl;

// out[0]: LUT (c=2); ival[0]: number of samples
extern "C" int ref_ibl_brdf_lut(const ref_args* a)
{
    const ref_img& o = a->out[0];
    const unsigned n = unsigned(a->ival[0]);
#pragma omp parallel for schedule(dynamic, 1)
    for (int y = 0; y < o.h; ++y)
        for (int x = 0; x < o.w; ++x)
        {
            float2 UV((float(x) + 0.5f) / float(o.w), (float(y) + 0.5f) / float(o.h));
            ref_store(o, x, y, i1::IntegrateBRDF(UV.y, UV.x, n));
        }
    return 0;
}

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

// in[0]: environment cube (mips); out[0]: one mip of the prefiltered cube (w x 6w, c=4); fval[0]: roughness; ival[0]: samples
extern "C" int ref_ibl_prefilter_env_map(const ref_args* a)
{
    ref_bind_cube(i2::g_EnvironmentMap.s, a, 0);
    i2::g_Roughness    = a->fval[0];
    i2::g_EnvMapWidth  = float(a->in[0][0].w);
    i2::g_EnvMapHeight = float(a->in[0][0].w);
    i2::g_EnvMipCount  = float(a->in_mips[0]);
    i2::g_NumSamples   = unsigned(a->ival[0]);
    const ref_img& o = a->out[0];
    for_each_cube_texel(o, [&](int x, int row, const float3& d) { ref_store(o, x, row, float4(i2::PrefilterEnvMap(i2::g_Roughness, d), 0.0f)); });
    return 0;
}

// in[0]: environment cube (mips); out[0]: irradiance cube (w x 6w, c=4); ival[0]: samples
extern "C" int ref_ibl_irradiance_map(const ref_args* a)
{
    ref_bind_cube(i3::g_EnvironmentMap.s, a, 0);
    i3::g_EnvMapWidth  = float(a->in[0][0].w);
    i3::g_EnvMapHeight = float(a->in[0][0].w);
    i3::g_EnvMipCount  = float(a->in_mips[0]);
    i3::g_NumSamples   = unsigned(a->ival[0]);
    const ref_img& o = a->out[0];
    for_each_cube_texel(o, [&](int x, int row, const float3& d) { ref_store(o, x, row, float4(i3::IrradianceMap(d), 1.0f)); });
    return 0;
}
