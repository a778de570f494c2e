// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D3: DOF_ComputeSeparatedCircleOfConfusion.fx (ComputeSeparatedCoCPS :5), host DepthOfField.cpp:879-903.
#include "ref_common.h"
namespace hlsl { namespace d3 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputeSeparatedCircleOfConfusion.fx"
}}
using namespace hlsl;

// in[0]: signed CoC; out[0]: near-field CoC magnitude (dilation mip 0)
extern "C" int ref_dof_separated_coc(const ref_args* a)
{
    ref_bind(d3::g_TextureCoC.s, a, 0);
    const ref_img& o = a->out[0];
    ref_fullscreen<d3::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](d3::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, d3::ComputeSeparatedCoCPS(vs)); });
    return 0;
}
