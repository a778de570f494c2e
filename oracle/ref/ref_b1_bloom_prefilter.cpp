// TEST INFRASTRUCTURE ONLY (oracle/_ref).  B1: Bloom_ComputePrefilteredTexture.fx (ComputePrefilteredTexturePS :37), host Bloom.cpp:288-311;
// input sampled with the linear BORDER sampler (Bloom.cpp:52-59,185; border colour 0).
#include "ref_common.h"
namespace hlsl { namespace b1 {
#include "ShaderDefinitions.fxh"
#include "BasicStructures.fxh"
#include "Bloom_ComputePrefilteredTexture.fx"
}}
using namespace hlsl;

// in[0]: colour (c=4); attribs: BloomAttribs; out[0]: half-resolution prefiltered colour (c=4, alpha 0)
extern "C" int ref_bloom_prefilter(const ref_args* a)
{
    ref_bind(b1::g_TextureInput.s, a, 0);
    b1::g_TextureInput_sampler = Sam_LinearBorder;
    std::memcpy(&b1::g_BloomAttribs, a->attribs, sizeof(b1::BloomAttribs));
    const ref_img& o = a->out[0];
    ref_fullscreen<b1::FullScreenTriangleVSOutput>(o.w, o.h, 0u, [&](b1::FullScreenTriangleVSOutput& vs, int x, int y) { ref_store(o, x, y, float4(b1::ComputePrefilteredTexturePS(vs), 0.0f)); });
    return 0;
}
extern "C" int ref_sizeof_bloom_attribs() { return int(sizeof(b1::BloomAttribs)); }
