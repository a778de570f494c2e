// ref_common.h -- TEST INFRASTRUCTURE ONLY.  Shared plumbing of the oracle/_ref pass wrappers:
// the C argument block, texture binding and the full-screen "rasterizer" loop that stands in for
// FullScreenTriangleVS (Shaders/Common/private/FullScreenTriangleVS.fx:3-28): one pixel-shader
// invocation per output texel at SV_Position = (x+0.5, y+0.5), f2NormalizedXY = NDC of the texel centre.
#pragma once
#include "hlsl_shim.h"

#include "../oracle_args.h"

namespace hlsl
{
inline void ref_bind(TexStorage& s, const ref_args* a, int slot)
{
    s.mips = a->in_mips[slot];
    for (int m = 0; m < s.mips; ++m)
    {
        s.mip[m].data = a->in[slot][m].data;
        s.mip[m].w    = a->in[slot][m].w;
        s.mip[m].h    = a->in[slot][m].h;
        s.mip[m].c    = a->in[slot][m].c;
    }
}
inline void ref_bind_cube(CubeStorage& s, const ref_args* a, int slot)
{
    s.mips = a->in_mips[slot];
    for (int m = 0; m < s.mips; ++m)
    {
        s.mip[m].data = a->in[slot][m].data;
        s.mip[m].w    = a->in[slot][m].w;
        s.mip[m].h    = a->in[slot][m].h;
        s.mip[m].c    = a->in[slot][m].c;
    }
}
inline void ref_store(const ref_img& o, int x, int y, float v) { o.data[(size_t(y) * o.w + x) * o.c] = v; }
inline void ref_store(const ref_img& o, int x, int y, const float2& v)
{
    float* p = o.data + (size_t(y) * o.w + x) * o.c;
    p[0] = v.x; p[1] = v.y;
}
inline void ref_store(const ref_img& o, int x, int y, const float3& v)
{
    float* p = o.data + (size_t(y) * o.w + x) * o.c;
    p[0] = v.x; p[1] = v.y; p[2] = v.z;
}
inline void ref_store(const ref_img& o, int x, int y, const float4& v)
{
    float* p = o.data + (size_t(y) * o.w + x) * o.c;
    for (int k = 0; k < o.c && k < 4; ++k) p[k] = v.d[k];
}

// Runs `ps(VSOut, x, y)` for every texel of a W x H target. `ps` returns false when the invocation discarded.
// VSOUT is the (per-namespace) FullScreenTriangleVSOutput type.
template <class VSOUT, class F> inline void ref_fullscreen(int W, int H, unsigned inst, F&& ps)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
        {
            VSOUT vs;
            vs.f4PixelPos     = float4(float(x) + 0.5f, float(y) + 0.5f, 0.0f, 1.0f);
            float u           = (float(x) + 0.5f) / float(W);
            float v           = (float(y) + 0.5f) / float(H);
            vs.f2NormalizedXY = float2(2.0f * u - 1.0f, 1.0f - 2.0f * v);
            vs.uInstID        = inst;
            g_ctx.discarded   = false;
            g_ctx.quad_phase  = -1;
            ps(vs, x, y);
        }
}
} // namespace hlsl
