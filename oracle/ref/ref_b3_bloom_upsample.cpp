// TEST INFRASTRUCTURE ONLY (oracle/_ref).  B3: Bloom_ComputeUpsampledTexture.fx (ComputeUpsampledTexturePS :20), host Bloom.cpp:339-396;
// both inputs linear CLAMP (Bloom.cpp:253-254); uInstID != 0 selects the final composite (:44-48, Draw with FirstInstance 3 at Bloom.cpp:383-392).
#include "ref_common.h"
namespace hlsl { namespace b3 {
#include "ShaderDefinitions.fxh"
#include "BasicStructures.fxh"
#include "PostFX_Common.fxh"
#include "Bloom_ComputeUpsampledTexture.fx"
}}
using namespace hlsl;

// in[0]: g_TextureInput (c=4), in[1]: g_TextureDownsampled (c=4); attribs; ival[0]: uInstID; out[0] (c=4; alpha: 0 for pyramid levels, input alpha for the final pass)
extern "C" int ref_bloom_upsample(const ref_args* a)
{
    ref_bind(b3::g_TextureInput.s, a, 0);
    ref_bind(b3::g_TextureDownsampled.s, a, 1);
    b3::g_TextureInput_sampler = b3::g_TextureDownsampled_sampler = Sam_LinearClamp;
    std::memcpy(&b3::g_BloomAttribs, a->attribs, sizeof(b3::BloomAttribs));
    const ref_img& o  = a->out[0];
    const ref_img& in = a->in[0][0];
    const bool final_pass = a->ival[0] != 0;
    ref_fullscreen<b3::FullScreenTriangleVSOutput>(o.w, o.h, unsigned(a->ival[0]), [&](b3::FullScreenTriangleVSOutput& vs, int x, int y) {
        float alpha = final_pass ? in.data[(size_t(y) * in.w + x) * in.c + 3] : 0.0f;
        ref_store(o, x, y, float4(b3::ComputeUpsampledTexturePS(vs), alpha));
    });
    return 0;
}
