// TEST INFRASTRUCTURE ONLY (oracle/_ref).  A6: SSAO_ComputeConvolutedDepthHistory.fx (ComputeConvolutedDepthHistoryPS :93),
// host: ScreenSpaceAmbientOcclusion.cpp:1075-1255 (SRV path; mip 0 of both pyramids are copies :1089-1106).
#include "ref_common.h"
#define SSAO_OPTION_INVERTED_DEPTH 0
#define SUPPORTED_SHADER_SRV 1
namespace hlsl { namespace a6 {
#include "ShaderDefinitions.fxh"
#include "SSAO_ComputeConvolutedDepthHistory.fx"
}}
using namespace hlsl;

// in[0]: AO history previous mip, in[1]: depth previous mip; out[0]: AO next mip, out[1]: depth next mip; ival[0]: mip index (uInstID)
extern "C" int ref_ssao_convoluted_history_mip(const ref_args* a)
{
    ref_bind(a6::g_TextureHistoryLastMip.s, a, 0);
    ref_bind(a6::g_TextureDepthLastMip.s, a, 1);
    const ref_img& o0 = a->out[0];
    const ref_img& o1 = a->out[1];
    ref_fullscreen<a6::FullScreenTriangleVSOutput>(o0.w, o0.h, unsigned(a->ival[0]), [&](a6::FullScreenTriangleVSOutput& vs, int x, int y) {
        a6::PSOutput r = a6::ComputeConvolutedDepthHistoryPS(vs);
        ref_store(o0, x, y, r.History);
        ref_store(o1, x, y, r.Depth);
    });
    return 0;
}
