// TEST INFRASTRUCTURE ONLY (oracle/_ref).  C3: Shaders/Common/private/ComputeClosestMotion.fx (ComputeClosestMotionPS :51),
// host wiring PostProcess/Common/src/PostFXContext.cpp:635-655.
#include "ref_common.h"
#ifndef POSTFX_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define POSTFX_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace c3 {
#include "ShaderDefinitions.fxh"
#include "ComputeClosestMotion.fx"
}}
using namespace hlsl;

// in[0]: depth, in[1]: motion (c=2); out[0]: closest motion (c=2)
extern "C" int ref_closest_motion(const ref_args* a)
{
    ref_bind(c3::g_TextureDepth.s, a, 0);
    ref_bind(c3::g_TextureMotion.s, a, 1);
    const ref_img& o = a->out[0];
    ref_fullscreen<c3::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](c3::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, c3::ComputeClosestMotionPS(vs)); });
    return 0;
}
