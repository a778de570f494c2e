// TEST INFRASTRUCTURE ONLY (oracle/_ref).  T0 / M2 host helpers: HaltonSequence + the arithmetic of GetJitterOffset
// (PostProcess/TemporalAntiAliasing/src/TemporalAntiAliasing.cpp:43-78), GetJitteredProjMatrix (.../interface/TemporalAntiAliasing.hpp:138-155) and
// ReverseExpToneMap (Components/src/ToneMapping.cpp:43-83), taken from the files where they lie (ref_prep.py EXTRACTS; the rest of those files needs DiligentCore).
// From DiligentCore (un-vendored submodule): Uint32 (Primitives/interface/BasicTypes.h) and the row-major Matrix4x4 with the named members m00 .. m33
// (Common/interface/BasicMath.hpp) -- restated here as far as the extracted code uses them.
#include "ref_common.h"
#include <algorithm>
#include <cmath>
namespace hlsl { namespace t0 {
using Uint32 = unsigned;
struct float4x4 // BasicMath.hpp Matrix4x4<float>: row-major, m<row><column>
{
    float m00, m01, m02, m03, m10, m11, m12, m13, m20, m21, m22, m23, m30, m31, m32, m33;
};
struct Holder // (GetJitteredProjMatrix is a static member function in the reference: give the keyword a class to live in)
{
#include "taa_host_extract.inc"
};
struct AccBufferT { Uint32 Width, Height, CurrentFrameIdx; };
static float2 jitter_offset(const AccBufferT& AccBuffer) // the tail of TemporalAntiAliasing::GetJitterOffset (:72-77) around the extracted statements
{
    auto HaltonSequence = [](Uint32 Base, Uint32 Index) { return Holder::HaltonSequence(Base, Index); };
#include "taa_jitter_statements_extract.inc"
    return float2{JitterX, JitterY};
}
#define constexpr const /* the shim's float3 has no constexpr constructor: `static constexpr float3 RGB_TO_LUMINANCE{...}` */
#include "tonemap_host_extract.inc"
#undef constexpr
}}
using namespace hlsl;

// out[0]: 1 x 1, c = 2: the jitter of frame ival[0] for a ival[1] x ival[2] accumulation buffer
extern "C" int ref_taa_jitter_offset(const ref_args* a)
{
    const ref_img& o = a->out[0];
    if (o.c != 2 || o.w < 1 || o.h < 1) return -1;
    const float2 j = t0::jitter_offset(t0::AccBufferT{unsigned(a->ival[1]), unsigned(a->ival[2]), unsigned(a->ival[0])});
    o.data[0] = j.x; o.data[1] = j.y;
    return 0;
}
// out[0]: 1 x 1, c = 1: HaltonSequence(ival[0], ival[1])
extern "C" int ref_halton_sequence(const ref_args* a)
{
    const ref_img& o = a->out[0];
    if (o.c != 1 || o.w < 1 || o.h < 1) return -1;
    o.data[0] = t0::Holder::HaltonSequence(unsigned(a->ival[0]), unsigned(a->ival[1]));
    return 0;
}
// in[0]: 4 x 4 (c = 1) projection, row-major; fval[0..1] = jitter; out[0]: 4 x 4
extern "C" int ref_taa_jittered_proj_matrix(const ref_args* a)
{
    const ref_img& i = a->in[0][0];
    const ref_img& o = a->out[0];
    if (i.c != 1 || i.w != 4 || i.h != 4 || o.c != 1 || o.w != 4 || o.h != 4) return -1;
    t0::float4x4 m;
    std::memcpy(&m, i.data, sizeof(m));
    const t0::float4x4 r = t0::Holder::GetJitteredProjMatrix(m, float2{a->fval[0], a->fval[1]});
    std::memcpy(o.data, &r, sizeof(r));
    return 0;
}
// in[0]: n x 1, c = 3 LDR colours; fval[0] = MiddleGray, fval[1] = AverageLogLum; out[0]: n x 1, c = 3
extern "C" int ref_reverse_exp_tone_map(const ref_args* a)
{
    const ref_img& i = a->in[0][0];
    const ref_img& o = a->out[0];
    if (i.c != 3 || o.c != 3 || i.w != o.w || i.h != o.h) return -1;
    for (size_t k = 0; k < size_t(i.w) * i.h; ++k)
    {
        const float3 r = t0::ReverseExpToneMap(float3{i.data[3 * k], i.data[3 * k + 1], i.data[3 * k + 2]}, a->fval[0], a->fval[1]);
        o.data[3 * k] = r.x; o.data[3 * k + 1] = r.y; o.data[3 * k + 2] = r.z;
    }
    return 0;
}
