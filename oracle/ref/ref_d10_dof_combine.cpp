// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D10: DOF_ComputeCombinedTexture.fx (ComputeCombinedTexturePS :36), host DepthOfField.cpp:1084-1114; linear CLAMP (:773-774).
#include "ref_common.h"
namespace hlsl { namespace d10 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputeCombinedTexture.fx"
}}
using namespace hlsl;

// in: 0 colour (full resolution), 1 CoC (bound, not read by the shader), 2 near, 3 far (half resolution); cam0; attribs; out[0]: rgb, a = the colour's alpha
// (the reference target is R11G11B10_FLOAT, DepthOfField.cpp:282-291; the fp32 plane carries the input alpha along, like the Bloom output)
extern "C" int ref_dof_combine(const ref_args* a)
{
    ref_bind(d10::g_TextureColor.s, a, 0);
    ref_bind(d10::g_TextureCoC.s, a, 1);
    ref_bind(d10::g_TextureDoFNearPlane.s, a, 2);
    ref_bind(d10::g_TextureDoFFarPlane.s, a, 3);
    d10::g_TextureDoFNearPlane_sampler = d10::g_TextureDoFFarPlane_sampler = Sam_LinearClamp;
    std::memcpy(&d10::g_Camera, a->cam0, sizeof(d10::CameraAttribs));
    std::memcpy(&d10::g_DOFAttribs, a->attribs, sizeof(d10::DepthOfFieldAttribs));
    const ref_img& o  = a->out[0];
    const ref_img& in = a->in[0][0];
    ref_fullscreen<d10::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](d10::FullScreenTriangleVSOutput& vs, int x, int y) {
        ref_store(o, x, y, float4(d10::ComputeCombinedTexturePS(vs), in.c > 3 ? in.data[(size_t(y) * in.w + x) * in.c + 3] : 1.0f));
    });
    return 0;
}
