// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R5: SSR_ComputeSpatialReconstruction.fx (ComputeSpatialReconstructionPS :114),
// host ScreenSpaceReflection.cpp:1001-1031 (masked; targets not cleared in the reference -- masked-out texels are 0 by contract here).
#include "ref_common.h"
#define SSR_OPTION_INVERTED_DEPTH 0
#ifndef SSR_OPTION_HALF_RESOLUTION // ref_r5_ssr_spatial_half.cpp builds the half-resolution permutation
#define SSR_OPTION_HALF_RESOLUTION 0
#endif
namespace hlsl { namespace r5 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeSpatialReconstruction.fx"
}}
using namespace hlsl;

// in: 0 roughness, 1 normal (c=4), 2 depth, 3 ray direction+PDF (c=4), 4 intersect specular (c=4), 5 mask; cam0; attribs
// out: 0 resolved radiance (c=4), 1 resolved variance, 2 resolved depth   (pre-filled with 0)
extern "C" int ref_ssr_spatial_reconstruction(const ref_args* a)
{
    ref_bind(r5::g_TextureRoughness.s, a, 0);
    ref_bind(r5::g_TextureNormal.s, a, 1);
    ref_bind(r5::g_TextureDepth.s, a, 2);
    ref_bind(r5::g_TextureRayDirectionPDF.s, a, 3);
    ref_bind(r5::g_TextureIntersectSpecular.s, a, 4);
    const ref_img& mask = a->in[5][0];
    std::memcpy(&r5::g_Camera, a->cam0, sizeof(r5::CameraAttribs));
    std::memcpy(&r5::g_SSRAttribs, a->attribs, sizeof(r5::ScreenSpaceReflectionAttribs));
    const ref_img &o0 = a->out[0], &o1 = a->out[1], &o2 = a->out[2];
    ref_fullscreen<r5::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](r5::FullScreenTriangleVSOutput& vs, int x, int y) {
        if (mask.data[size_t(y) * mask.w + x] == 0.0f) return;
        r5::PSOutput r = r5::ComputeSpatialReconstructionPS(vs);
        ref_store(o0, x, y, r.ResolvedRadiance);
        ref_store(o1, x, y, r.ResolvedVariance);
        ref_store(o2, x, y, r.ResolvedDepth);
    });
    return 0;
}
