// ref_tonemap.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref).
// Compiles the reference's ToneMap() (Shaders/PostProcess/ToneMapping/public/ToneMapping.fxh:87-226)
// once per TONE_MAPPING_MODE and applies it full-screen exactly as the copy-frame pass does
// (Hydrogent/shaders/HnCopyFrame.psh:27-36,61-63: Color = Load; Color.rgb = ToneMap(...); [LinearToSRGB]).
#include "ref_common.h"

namespace hlsl
{
#include "ShaderDefinitions.fxh"
#include "ToneMappingStructures.fxh"
#include "SRGBUtilities.fxh"

#define REF_TM_NS(N)
namespace tm0
{
#define TONE_MAPPING_MODE 0
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm1
{
#define TONE_MAPPING_MODE 1
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm2
{
#define TONE_MAPPING_MODE 2
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm3
{
#define TONE_MAPPING_MODE 3
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm4
{
#define TONE_MAPPING_MODE 4
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm5
{
#define TONE_MAPPING_MODE 5
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm6
{
#define TONE_MAPPING_MODE 6
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm7
{
#define TONE_MAPPING_MODE 7
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm8
{
#define TONE_MAPPING_MODE 8
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm9
{
#define TONE_MAPPING_MODE 9
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm10
{
#define TONE_MAPPING_MODE 10
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}
namespace tm11
{
#define TONE_MAPPING_MODE 11
#include "ToneMapping.fxh"
#undef TONE_MAPPING_MODE
}

typedef float3 (*tonemap_fn)(float3, ToneMappingAttribs, float);
static const tonemap_fn k_tonemap[12] = {tm0::ToneMap, tm1::ToneMap, tm2::ToneMap,  tm3::ToneMap, tm4::ToneMap,  tm5::ToneMap,
                                         tm6::ToneMap, tm7::ToneMap, tm8::ToneMap,  tm9::ToneMap, tm10::ToneMap, tm11::ToneMap};
} // namespace hlsl

using namespace hlsl;

// in[0]: HDR colour (c = 4); out[0]: colour (c = 4); attribs: ToneMappingAttribs; fval[0]: fAveLogLum;
// ival[0]: 1 = convert the result to sRGB (CONVERT_OUTPUT_TO_SRGB)
extern "C" int ref_tonemap(const ref_args* a)
{
    ToneMappingAttribs attr;
    std::memcpy(&attr, a->attribs, sizeof(attr));
    int mode = attr.iToneMappingMode;
    if (mode < 0 || mode > 11) return -1;
    tonemap_fn   fn   = k_tonemap[mode];
    const float  lum  = a->fval[0];
    const bool   srgb = a->ival[0] != 0;
    const ref_img& in = a->in[0][0];
    const ref_img& out = a->out[0];
#pragma omp parallel for
    for (int y = 0; y < in.h; ++y)
        for (int x = 0; x < in.w; ++x)
        {
            const float* p = in.data + (size_t(y) * in.w + x) * in.c;
            float4 c(p[0], p[1], p[2], p[3]);
            float3 t = fn(float3(c.x, c.y, c.z), attr, lum);
            if (srgb) t = LinearToSRGB(t);
            ref_store(out, x, y, float4(t, c.w));
        }
    return 0;
}

extern "C" int ref_sizeof_tonemap_attribs() { return int(sizeof(ToneMappingAttribs)); }
