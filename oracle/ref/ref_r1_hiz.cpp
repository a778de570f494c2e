// TEST INFRASTRUCTURE ONLY (oracle/_ref).  R1: SSR_ComputeHierarchicalDepthBuffer.fx (ComputeHierarchicalDepthBufferPS :30),
// host: ScreenSpaceReflection.cpp:777-902 (SRV path; mip 0 = copy of the depth :789-806).
#include "ref_common.h"
#ifndef SSR_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define SSR_OPTION_INVERTED_DEPTH 0
#endif
#define SUPPORTED_SHADER_SRV 1
namespace hlsl { namespace r1 {
#include "ShaderDefinitions.fxh"
#include "SSR_ComputeHierarchicalDepthBuffer.fx"
}}
using namespace hlsl;

// in[0]: previous mip; out[0]: next mip; ival[0]: mip index
extern "C" int ref_ssr_hiz_mip(const ref_args* a)
{
    ref_bind(r1::g_TextureLastMip.s, a, 0);
    const ref_img& o = a->out[0];
    ref_fullscreen<r1::FullScreenTriangleVSOutput>(o.w, o.h, unsigned(a->ival[0]), [&](r1::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, r1::ComputeHierarchicalDepthBufferPS(vs)); });
    return 0;
}
