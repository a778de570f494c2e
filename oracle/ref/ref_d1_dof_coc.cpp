// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D1: DOF_ComputeCircleOfConfusion.fx (ComputeCircleOfConfusionPS :24), host DepthOfField.cpp:820-847.
#include "ref_common.h"
namespace hlsl { namespace d1 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputeCircleOfConfusion.fx"
}}
using namespace hlsl;

// in[0]: depth (c=1); cam0; attribs = DepthOfFieldAttribs; out[0]: signed CoC (c=1)
extern "C" int ref_dof_coc(const ref_args* a)
{
    ref_bind(d1::g_TextureDepth.s, a, 0);
    std::memcpy(&d1::g_Camera, a->cam0, sizeof(d1::CameraAttribs));
    std::memcpy(&d1::g_DOFAttribs, a->attribs, sizeof(d1::DepthOfFieldAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<d1::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](d1::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, d1::ComputeCircleOfConfusionPS(vs)); });
    return 0;
}
