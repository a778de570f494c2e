// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A1: SSAO_ComputeDownsampledDepth.fx (ComputeDownsampledDepthPS :24), host ScreenSpaceAmbientOcclusion.cpp:818-838
// (FEATURE_FLAG_HALF_RESOLUTION only): checkerboard of the min / max depth of every 2x2 block.
#include "ref_common.h"
#define SSAO_OPTION_INVERTED_DEPTH 0
namespace hlsl { namespace a1 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputeDownsampledDepth.fx"
}}
using namespace hlsl;

// in[0]: depth (full resolution); out[0]: checkerboard depth ((W / 2) x (H / 2))
extern "C" int ref_ssao_downsampled_depth(const ref_args* a)
{
    ref_bind(a1::g_TextureDepth.s, a, 0);
    const ref_img& o = a->out[0];
    ref_fullscreen<a1::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](a1::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, a1::ComputeDownsampledDepthPS(vs)); });
    return 0;
}
