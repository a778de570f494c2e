// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D6: DOF_ComputePrefilteredTexture.fx (ComputePrefilteredTexturePS :23), host DepthOfField.cpp:973-1006;
// g_TextureDilationCoC linear CLAMP (:614); targets are (W/2) x (H/2) (:256-266).
#include "ref_common.h"
namespace hlsl { namespace d6 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputePrefilteredTexture.fx"
}}
using namespace hlsl;

// in: 0 colour, 1 signed CoC (full resolution), 2 blurred dilation CoC (last level); attribs; out: 0 near (rgb, dilated near CoC), 1 far (rgb, far CoC)
extern "C" int ref_dof_prefilter(const ref_args* a)
{
    ref_bind(d6::g_TextureColor.s, a, 0);
    ref_bind(d6::g_TextureCoC.s, a, 1);
    ref_bind(d6::g_TextureDilationCoC.s, a, 2);
    d6::g_TextureDilationCoC_sampler = Sam_LinearClamp;
    std::memcpy(&d6::g_DOFAttribs, a->attribs, sizeof(d6::DepthOfFieldAttribs));
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<d6::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](d6::FullScreenTriangleVSOutput& vs, int x, int y) {
        d6::PSOutput r = d6::ComputePrefilteredTexturePS(vs);
        ref_store(o0, x, y, r.ForegroundColor);
        ref_store(o1, x, y, r.BackgroundColor);
    });
    return 0;
}
