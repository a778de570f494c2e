// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A8: SSAO_ComputeSpatialReconstruction.fx (ComputeSpatialReconstructionPS :49),
// host: ScreenSpaceAmbientOcclusion.cpp:1288-1329 (the resolved AO is then copied into the current history slot :1319-1328).
#include "ref_common.h"
#ifndef SSAO_OPTION_INVERTED_DEPTH // ref_*_rev.cpp builds the reversed-depth permutation of this file
#define SSAO_OPTION_INVERTED_DEPTH 0
#endif
namespace hlsl { namespace a8 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputeSpatialReconstruction.fx"
}}
using namespace hlsl;

// in: 0 resampled AO, 1 history length, 2 depth, 3 normal (c=4); cam0; attribs; out[0]: AO
extern "C" int ref_ssao_spatial_reconstruction(const ref_args* a)
{
    ref_bind(a8::g_TextureOcclusion.s, a, 0);
    ref_bind(a8::g_TextureHistory.s, a, 1);
    ref_bind(a8::g_TextureDepth.s, a, 2);
    ref_bind(a8::g_TextureNormal.s, a, 3);
    std::memcpy(&a8::g_Camera, a->cam0, sizeof(a8::CameraAttribs));
    std::memcpy(&a8::g_SSAOAttribs, a->attribs, sizeof(a8::ScreenSpaceAmbientOcclusionAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<a8::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](a8::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, a8::ComputeSpatialReconstructionPS(vs)); });
    return 0;
}
