// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D2: DOF_ComputeTemporalCircleOfConfusion.fx (ComputeTemporalCircleOfConfusionPS :74), host DepthOfField.cpp:848-878:
// g_TexturePrevCoC linear CLAMP, g_TextureCurrCoC point CLAMP (:445-446); g_TextureMotion = PostFX closest motion (:872).
#include "ref_common.h"
namespace hlsl { namespace d2 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputeTemporalCircleOfConfusion.fx"
}}
using namespace hlsl;

// in: 0 current CoC, 1 previous temporal CoC, 2 closest motion (c=2); cam0; attribs; out[0]: temporal CoC
extern "C" int ref_dof_temporal_coc(const ref_args* a)
{
    ref_bind(d2::g_TextureCurrCoC.s, a, 0);
    ref_bind(d2::g_TexturePrevCoC.s, a, 1);
    ref_bind(d2::g_TextureMotion.s, a, 2);
    d2::g_TexturePrevCoC_sampler = Sam_LinearClamp;
    d2::g_TextureCurrCoC_sampler = Sam_PointClamp;
    std::memcpy(&d2::g_Camera, a->cam0, sizeof(d2::CameraAttribs));
    std::memcpy(&d2::g_DOFAttribs, a->attribs, sizeof(d2::DepthOfFieldAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<d2::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](d2::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, d2::ComputeTemporalCircleOfConfusionPS(vs)); });
    return 0;
}
