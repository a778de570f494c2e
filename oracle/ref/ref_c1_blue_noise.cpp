// TEST INFRASTRUCTURE ONLY (oracle/_ref).  C1: Shaders/Common/private/ComputeBlueNoiseTexture.fx (ComputeBlueNoiseTexturePS :81),
// host wiring PostProcess/Common/src/PostFXContext.cpp:567-607 (frame index travels as uInstID; RG8_UNORM targets :200).
#include "ref_common.h"
namespace hlsl { namespace c1 {
#include "ShaderDefinitions.fxh"
#include "ComputeBlueNoiseTexture.fx"
}}
using namespace hlsl;

static inline float unorm8(float v) // RG8_UNORM render target: store + load
{
    v = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
    return std::floor(v * 255.0f + 0.5f) / 255.0f;
}

// in[0]: Sobol 256x1 (c=1, byte values as floats), in[1]: scrambling tile 512x256 (c=1); out[0]: XY 128x128 c=2, out[1]: ZW; ival[0] = frame index
extern "C" int ref_blue_noise(const ref_args* a)
{
    ref_bind(c1::g_SobolBuffer.s, a, 0);
    ref_bind(c1::g_ScramblingTileBuffer.s, a, 1);
    const ref_img& oxy = a->out[0];
    const ref_img& ozw = a->out[1];
    ref_fullscreen<c1::FullScreenTriangleVSOutput>(128, 128, unsigned(a->ival[0]), [&](c1::FullScreenTriangleVSOutput& vs, int x, int y) {
        c1::PSOutput o = c1::ComputeBlueNoiseTexturePS(vs);
        ref_store(oxy, x, y, float2(unorm8(o.BlueNoiseXY.x), unorm8(o.BlueNoiseXY.y)));
        ref_store(ozw, x, y, float2(unorm8(o.BlueNoiseZW.x), unorm8(o.BlueNoiseZW.y)));
    });
    return 0;
}
