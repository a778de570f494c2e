// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D9: DOF_ComputePostfilteredTexture.fx (ComputePostfilteredTexturePS :26), host DepthOfField.cpp:1062-1083; linear CLAMP (:734-735).
#include "ref_common.h"
namespace hlsl { namespace d9 {
#include "ShaderDefinitions.fxh"
#include "BasicStructures.fxh"
#include "PostFX_Common.fxh"
#include "DOF_ComputePostfilteredTexture.fx"
}}
using namespace hlsl;

// in: 0 near, 1 far (second bokeh pass); out: 0 near, 1 far (2x2 tent)
extern "C" int ref_dof_postfilter(const ref_args* a)
{
    ref_bind(d9::g_TextureColorCoCNear.s, a, 0);
    ref_bind(d9::g_TextureColorCoCFar.s, a, 1);
    d9::g_TextureColorCoCNear_sampler = d9::g_TextureColorCoCFar_sampler = Sam_LinearClamp;
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<d9::FullScreenTriangleVSOutput>(o0.w, o0.h, 0u, [&](d9::FullScreenTriangleVSOutput& vs, int x, int y) {
        d9::PSOutput r = d9::ComputePostfilteredTexturePS(vs);
        ref_store(o0, x, y, r.ForegroundColor);
        ref_store(o1, x, y, r.BackgroundColor);
    });
    return 0;
}
