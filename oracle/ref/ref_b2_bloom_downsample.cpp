// TEST INFRASTRUCTURE ONLY (oracle/_ref).  B2: Bloom_ComputeDownsampledTexture.fx (ComputeDownsampledTexturePS :11), host Bloom.cpp:313-337;
// linear BORDER sampler (Bloom.cpp:219).
#include "ref_common.h"
namespace hlsl { namespace b2 {
#include "ShaderDefinitions.fxh"
#include "BasicStructures.fxh"
#include "PostFX_Common.fxh"
#include "Bloom_ComputeDownsampledTexture.fx"
}}
using namespace hlsl;

// in[0]: level i-1 (c=4); out[0]: level i (c=4)
extern "C" int ref_bloom_downsample(const ref_args* a)
{
    ref_bind(b2::g_TextureInput.s, a, 0);
    b2::g_TextureInput_sampler = Sam_LinearBorder;
    const ref_img& o = a->out[0];
    ref_fullscreen<b2::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](b2::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, float4(b2::ComputeDownsampledTexturePS(vs), 0.0f)); });
    return 0;
}
