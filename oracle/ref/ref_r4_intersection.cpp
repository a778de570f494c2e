// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R4: SSR_ComputeIntersection.fx (ComputeIntersectionPS :281), host ScreenSpaceReflection.cpp:963-999:
// blue noise XY (:977), Hi-Z via Load, both targets cleared to 0 (:993-994), executed only under the mask (depth test LESS vs the D16 mask).
#include "ref_common.h"
#define SSR_OPTION_INVERTED_DEPTH 0
#define SSR_OPTION_PREVIOUS_FRAME 0
#define SSR_OPTION_HALF_RESOLUTION 0
namespace hlsl { namespace r4 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeIntersection.fx"
}}
using namespace hlsl;

// in: 0 radiance (c=4), 1 normal (c=4), 2 roughness, 3 blue noise XY (c=2), 4 Hi-Z (7 mips), 5 mask; cam0; attribs
// out: 0 specular (radiance, confidence), 1 direction*length + PDF   (both pre-filled with 0)
extern "C" int ref_ssr_intersection(const ref_args* a)
{
    ref_bind(r4::g_TextureRadiance.s, a, 0);
    ref_bind(r4::g_TextureNormal.s, a, 1);
    ref_bind(r4::g_TextureRoughness.s, a, 2);
    ref_bind(r4::g_TextureBlueNoise.s, a, 3);
    ref_bind(r4::g_TextureDepthHierarchy.s, a, 4);
    const ref_img& mask = a->in[5][0];
    std::memcpy(&r4::g_Camera, a->cam0, sizeof(r4::CameraAttribs));
    std::memcpy(&r4::g_SSRAttribs, a->attribs, sizeof(r4::ScreenSpaceReflectionAttribs));
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<r4::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](r4::FullScreenTriangleVSOutput& vs, int x, int y) {
        if (mask.data[size_t(y) * mask.w + x] == 0.0f) return;
        r4::PSOutput r = r4::ComputeIntersectionPS(vs);
        ref_store(o0, x, y, r.Specular);
        ref_store(o1, x, y, r.DirectionPDF);
    });
    return 0;
}
