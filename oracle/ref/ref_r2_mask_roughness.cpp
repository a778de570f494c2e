// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R2: SSR_ComputeStencilMaskAndExtractRoughness.fx (:32), host ScreenSpaceReflection.cpp:904-932.
// The reference writes the roughness only where the pixel is a reflection sample (the RT is not cleared, so other texels hold stale
// data) and marks those pixels in a D16 depth mask.  Contract here: out[0] = LoadRoughness() for EVERY texel,
// out[1] = mask (1.0 where the shader did not discard).
#include "ref_common.h"
#ifndef SSR_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define SSR_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace r2 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeStencilMaskAndExtractRoughness.fx"
}}
using namespace hlsl;

// in[0]: material (c=4), in[1]: depth; attribs: ScreenSpaceReflectionAttribs; out[0]: roughness, out[1]: mask
extern "C" int ref_ssr_mask_roughness(const ref_args* a)
{
    ref_bind(r2::g_TextureMaterialParameters.s, a, 0);
    ref_bind(r2::g_TextureDepth.s, a, 1);
    std::memcpy(&r2::g_SSRAttribs, a->attribs, sizeof(r2::ScreenSpaceReflectionAttribs));
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<r2::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](r2::FullScreenTriangleVSOutput& vs, int x, int y) {
        float r = r2::ComputeStencilMaskAndExtractRoughnessPS(vs);
        bool kept = !g_ctx.discarded;
        ref_store(o0, x, y, kept ? r : r2::LoadRoughness(int2(x, y)));
        ref_store(o1, x, y, kept ? 1.0f : 0.0f);
    });
    return 0;
}
extern "C" int ref_sizeof_ssr_attribs() { return int(sizeof(r2::ScreenSpaceReflectionAttribs)); }
