// TEST INFRASTRUCTURE ONLY (oracle/_ref).  D4: DOF_ComputeDilationCircleOfConfusion.fx (ComputeDilationCoCPS :14), host DepthOfField.cpp:904-925
// (one draw per dilation level; level k is (W >> k) x (H >> k), DepthOfField.cpp:231-241).
#include "ref_common.h"
namespace hlsl { namespace d4 {
#include "ShaderDefinitions.fxh"
#include "DOF_ComputeDilationCircleOfConfusion.fx"
}}
using namespace hlsl;

// in[0]: previous dilation level; out[0]: next level (max over the 2x2 / 3x3 footprint)
extern "C" int ref_dof_dilation_coc(const ref_args* a)
{
    ref_bind(d4::g_TextureLastMip.s, a, 0);
    const ref_img& o = a->out[0];
    ref_fullscreen<d4::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](d4::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, d4::ComputeDilationCoCPS(vs)); });
    return 0;
}
