// ref_n3_autoexposure.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
// Auto exposure from the reference's own text:
//   * RGB_TO_LUMINANCE, GetWeightedLogLum    Shaders/PostProcess/EpipolarLightScattering/private/AtmosphereShadersCommon.fxh:84,197-203
//   * UpdateAverageLuminancePS               .../private/UpdateAverageLuminance.fx:12-29
// ref_prep.py extracts exactly these definitions from the files where they lie (the rest of AtmosphereShadersCommon.fxh drags in the whole
// light-scattering post-process) into epls_autoexposure_extract.inc in the temporary build directory.
// Host side restated here: the 64x64 luminance target is rendered as UnwarpEpipolarScattering.fx:283-307 does with the scene colour only
// (no in-scattering, extinction 1, bShowLightingOnly off), IDeviceContext::GenerateMips is a 2x2 box filter per level, and the pixel
// shader's output is blended with BS_AlphaBlend (EpipolarLightScattering.cpp:1827, 2496-2506).
#include "ref_common.h"
#include <vector>

namespace hlsl
{
struct MiscDynamicParams // the one field UpdateAverageLuminancePS reads (EpipolarLightScatteringStructures.fxh)
{
    float fElapsedTime;
};
struct FullScreenTriangleVSOutput
{
    float4 f4PixelPos;
};
#define LOW_RES_LUMINANCE_MIPS 7
namespace adapt1
{
#define LIGHT_ADAPTATION 1
MiscDynamicParams  g_MiscParams;
Texture2D_<float2> g_tex2DLowResLuminance;
#include "epls_autoexposure_extract.inc"
#undef LIGHT_ADAPTATION
#undef RGB_TO_LUMINANCE
} // namespace adapt1
namespace adapt0
{
#define LIGHT_ADAPTATION 0
Texture2D_<float2> g_tex2DLowResLuminance;
#include "epls_autoexposure_extract.inc"
#undef LIGHT_ADAPTATION
#undef RGB_TO_LUMINANCE
} // namespace adapt0
} // namespace hlsl
using namespace hlsl;

// in[0]: scene colour (c=4); out[0]: low-res luminance 64x64 (c=2); out[1]: average luminance 1x1 (c=1, read-modify-write);
// fval[0]: elapsed time; ival[0]: LIGHT_ADAPTATION
extern "C" int ref_autoexposure(const ref_args* a)
{
    Texture2D_<float4> color;
    ref_bind(color.s, a, 0);
    const ref_img& low = a->out[0];
    if (low.w != 64 || low.h != 64 || low.c != 2) return -1;
    for (int y = 0; y < 64; ++y)
        for (int x = 0; x < 64; ++x)
        {
            const float2 uv  = float2((float(x) + 0.5f) / 64.0f, (float(y) + 0.5f) / 64.0f);
            const float3 rgb = color.SampleLevel(Sam_LinearClamp, uv, 0).rgb;
            ref_store(low, x, y, adapt1::GetWeightedLogLum(rgb, 0.01f));
        }
    // GenerateMips: every level is the 2x2 box filter of the previous one
    std::vector<std::vector<float>> chain;
    chain.emplace_back(low.data, low.data + 64 * 64 * 2);
    for (int n = 64; n > 1; n /= 2)
    {
        const std::vector<float>& src = chain.back();
        std::vector<float> dst(size_t(n / 2) * (n / 2) * 2);
        for (int y = 0; y < n / 2; ++y)
            for (int x = 0; x < n / 2; ++x)
                for (int c = 0; c < 2; ++c)
                {
                    auto at = [&](int xx, int yy) { return src[(size_t(yy) * n + xx) * 2 + c]; };
                    dst[(size_t(y) * (n / 2) + x) * 2 + c] = ((at(2 * x, 2 * y) + at(2 * x + 1, 2 * y)) + (at(2 * x, 2 * y + 1) + at(2 * x + 1, 2 * y + 1))) * 0.25f;
                }
        chain.push_back(std::move(dst));
    }
    TexStorage& st = a->ival[0] ? adapt1::g_tex2DLowResLuminance.s : adapt0::g_tex2DLowResLuminance.s;
    st.mips = int(chain.size());
    for (int m = 0, n = 64; m < st.mips; ++m, n /= 2)
    {
        st.mip[m].data = chain[size_t(m)].data();
        st.mip[m].w = st.mip[m].h = n;
        st.mip[m].c = 2;
    }
    FullScreenTriangleVSOutput vs{};
    float4 outLum;
    if (a->ival[0])
    {
        adapt1::g_MiscParams.fElapsedTime = a->fval[0];
        adapt1::UpdateAverageLuminancePS(vs, outLum);
    }
    else
        adapt0::UpdateAverageLuminancePS(vs, outLum);
    float* avg = a->out[1].data;
    *avg = outLum.x * outLum.w + *avg * (1.0f - outLum.w); // BS_AlphaBlend on the R channel
    return 0;
}
