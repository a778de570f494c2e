// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R3: SSR_ComputeDownsampledStencilMask.fx (ComputeDownsampledStencilMaskPS :24), host ScreenSpaceReflection.cpp:934-961
// (FEATURE_FLAG_HALF_RESOLUTION only): a half-resolution texel is marked when the closest depth / largest roughness of its 2x2 (3x3 at odd sizes) block is a reflection sample.
#include "ref_common.h"
#ifndef SSR_OPTION_INVERTED_DEPTH
#define SSR_OPTION_INVERTED_DEPTH 0
#endif
#undef discard // this pixel shader returns void (it only writes the stencil mask): the shim's value-returning discard does not apply
#define discard do { ::hlsl::g_ctx.discarded = true; return; } while (0)
namespace hlsl { namespace r3 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeDownsampledStencilMask.fx"
}}
using namespace hlsl;

// in: 0 roughness (R2), 1 depth; attribs; out[0]: mask (half resolution, pre-filled with 0; 1 where the shader did not discard)
extern "C" int ref_ssr_downsampled_mask(const ref_args* a)
{
    ref_bind(r3::g_TextureRoughness.s, a, 0);
    ref_bind(r3::g_TextureDepth.s, a, 1);
    std::memcpy(&r3::g_SSRAttribs, a->attribs, sizeof(r3::ScreenSpaceReflectionAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<r3::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](r3::FullScreenTriangleVSOutput& vs, int x, int y) {
        r3::ComputeDownsampledStencilMaskPS(vs);
        ref_store(o, x, y, g_ctx.discarded ? 0.0f : 1.0f);
    });
    return 0;
}
