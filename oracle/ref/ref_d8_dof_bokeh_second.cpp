// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D8: DOF_ComputeBokehSecondPass.fx (ComputeBokehPS :40), host DepthOfField.cpp:1035-1061; linear CLAMP (:699-700);
// kernel = GenerateKernelPoints(DOF_BOKEH_KERNEL_SMALL_RING_COUNT, DOF_BOKEH_KERNEL_SMALL_RING_DENSITY) (DepthOfField.cpp:131-150).
#include "ref_common.h"
namespace hlsl { namespace d8 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputeBokehSecondPass.fx"
}}
using namespace hlsl;

// in: 0 near, 1 far (first bokeh pass), 2 small kernel (c=2); cam0; attribs; out: 0 near, 1 far (flood fill: running max)
extern "C" int ref_dof_bokeh_second(const ref_args* a)
{
    ref_bind(d8::g_TextureColorCoCNear.s, a, 0);
    ref_bind(d8::g_TextureColorCoCFar.s, a, 1);
    ref_bind(d8::g_TextureBokehKernel.s, a, 2);
    d8::g_TextureColorCoCNear_sampler = d8::g_TextureColorCoCFar_sampler = Sam_LinearClamp;
    std::memcpy(&d8::g_Camera, a->cam0, sizeof(d8::CameraAttribs));
    std::memcpy(&d8::g_DOFAttribs, a->attribs, sizeof(d8::DepthOfFieldAttribs));
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<d8::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](d8::FullScreenTriangleVSOutput& vs, int x, int y) {
        d8::PSOutput r = d8::ComputeBokehPS(vs);
        ref_store(o0, x, y, r.ForegroundColor);
        ref_store(o1, x, y, r.BackgroundColor);
    });
    return 0;
}
