// oracle_kit.h -- TEST INFRASTRUCTURE ONLY.  Scalar C++ vocabulary of the hand-written CPU oracle: small vector types,
// camera layout, image/pyramid accessors with the texture-unit semantics the reference relies on:
// Load out of bounds -> 0, clamp addressing, bilinear weights of GetBilinearSamplingInfoUC (ShaderUtilities.fxh:126-142),
// point-mip selection floor(lod + 0.5), D3D min/max/saturate NaN rules.  No dependence on the product sources.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>

#include "oracle_args.h"

namespace ok
{
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };

#define OK_OPS2(op)                                                  \
    inline f2 operator op(f2 a, f2 b) { return {a.x op b.x, a.y op b.y}; } \
    inline f2 operator op(f2 a, float b) { return {a.x op b, a.y op b}; }  \
    inline f2 operator op(float a, f2 b) { return {a op b.x, a op b.y}; }
#define OK_OPS3(op)                                                              \
    inline f3 operator op(f3 a, f3 b) { return {a.x op b.x, a.y op b.y, a.z op b.z}; } \
    inline f3 operator op(f3 a, float b) { return {a.x op b, a.y op b, a.z op b}; }    \
    inline f3 operator op(float a, f3 b) { return {a op b.x, a op b.y, a op b.z}; }
#define OK_OPS4(op)                                                                          \
    inline f4 operator op(f4 a, f4 b) { return {a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w}; } \
    inline f4 operator op(f4 a, float b) { return {a.x op b, a.y op b, a.z op b, a.w op b}; }      \
    inline f4 operator op(float a, f4 b) { return {a op b.x, a op b.y, a op b.z, a op b.w}; }
OK_OPS2(+) OK_OPS2(-) OK_OPS2(*) OK_OPS2(/)
OK_OPS3(+) OK_OPS3(-) OK_OPS3(*) OK_OPS3(/)
OK_OPS4(+) OK_OPS4(-) OK_OPS4(*) OK_OPS4(/)
inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
inline f3& operator+=(f3& a, f3 b) { a = a + b; return a; }
inline f4& operator+=(f4& a, f4 b) { a = a + b; return a; }

// D3D / IEEE-754-2008 minNum/maxNum semantics: a NaN operand is ignored; saturate(NaN) = 0
inline float fmin2(float a, float b) { return std::fmin(a, b); }
inline float fmax2(float a, float b) { return std::fmax(a, b); }
inline float sat(float x) { return fmin2(fmax2(x, 0.0f), 1.0f); }
inline float clampf(float x, float a, float b) { return fmin2(fmax2(x, a), b); }
inline int   clampi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline f3    lerp(f3 a, f3 b, float t) { return a + t * (b - a); }
inline f3    lerp(f3 a, f3 b, f3 t) { return a + t * (b - a); }
inline f4    lerp(f4 a, f4 b, float t) { return a + t * (b - a); }
inline float frac(float x) { return x - std::floor(x); }
inline float sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
inline float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float length(f2 a) { return std::sqrt(dot(a, a)); }
inline float length(f3 a) { return std::sqrt(dot(a, a)); }
inline f3    normalize(f3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline f3    cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline f3    reflect(f3 i, f3 n) { return i - 2.0f * dot(n, i) * n; }
inline f3    splat3(float s) { return {s, s, s}; }
inline f4    splat4(float s) { return {s, s, s, s}; }
inline f3    xyz(f4 a) { return {a.x, a.y, a.z}; }
inline f4    mk4(f3 a, float w) { return {a.x, a.y, a.z, w}; }
inline f3    max3(f3 a, float b) { return {fmax2(a.x, b), fmax2(a.y, b), fmax2(a.z, b)}; }
inline f3    max3(f3 a, f3 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z)}; }
inline f4    max4(f4 a, float b) { return {fmax2(a.x, b), fmax2(a.y, b), fmax2(a.z, b), fmax2(a.w, b)}; }
inline f4    max4(f4 a, f4 b) { return {fmax2(a.x, b.x), fmax2(a.y, b.y), fmax2(a.z, b.z), fmax2(a.w, b.w)}; }
inline f4    min4(f4 a, f4 b) { return {fmin2(a.x, b.x), fmin2(a.y, b.y), fmin2(a.z, b.z), fmin2(a.w, b.w)}; }
inline f3    sqrt3(f3 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z)}; }
inline f4    sqrt4(f4 a) { return {std::sqrt(a.x), std::sqrt(a.y), std::sqrt(a.z), std::sqrt(a.w)}; }
inline float max_comp(f3 a) { return fmax2(a.x, fmax2(a.y, a.z)); }

// CameraAttribs -- Shaders/Common/public/BasicStructures.fxh:84-149
struct Camera
{
    float pos[4], viewport[4];
    float nearZ, farZ, nearDepth, farDepth, sceneNearZ, sceneFarZ, sceneNearDepth, sceneFarDepth;
    float handness; uint32_t frameIndex; float pad0, pad1;
    float focusDistance, fStop, focalLength, sensorWidth, sensorHeight, exposure, jitter[2];
    float view[16], proj[16], viewProj[16], viewInv[16], projInv[16], viewProjInv[16];
    float extra[20];
};
static_assert(sizeof(Camera) == 576, "CameraAttribs layout");
inline Camera load_camera(const void* p) { Camera c; std::memcpy(&c, p, sizeof(c)); return c; }

inline f4 mul(f4 v, const float* M) // row vector x row-major matrix
{
    return {v.x * M[0] + v.y * M[4] + v.z * M[8] + v.w * M[12], v.x * M[1] + v.y * M[5] + v.z * M[9] + v.w * M[13],
            v.x * M[2] + v.y * M[6] + v.z * M[10] + v.w * M[14], v.x * M[3] + v.y * M[7] + v.z * M[11] + v.w * M[15]};
}
inline f3 mul_dir(f3 d, const float* M) // mul(float4(d, 0), M).xyz
{
    return {d.x * M[0] + d.y * M[4] + d.z * M[8], d.x * M[1] + d.y * M[5] + d.z * M[9], d.x * M[2] + d.y * M[6] + d.z * M[10]};
}
// ShaderUtilities.fxh:5-40
inline float camera_z_to_depth(float z, const float* P) { return (P[10] * z + P[14]) / (P[11] * z + P[15]); }
inline float depth_to_camera_z(float d, const float* P) { return (P[14] - d * P[15]) / (d * P[11] - P[10]); }
// Appendix A (D3D / Vulkan)
inline f2 ndc_to_uv(f2 xy) { return {0.5f + 0.5f * xy.x, 0.5f + -0.5f * xy.y}; }
inline f2 uv_to_ndc(f2 uv) { return {(uv.x - 0.5f) * 2.0f, (uv.y - 0.5f) * -2.0f}; }
// PostFX_Common.fxh:85-111
inline f3 project_position(f3 o, const float* T)
{
    f4 p = mul({o.x, o.y, o.z, 1.0f}, T);
    f3 q = {p.x / p.w, p.y / p.w, p.z / p.w};
    f2 uv = ndc_to_uv({q.x, q.y});
    return {uv.x, uv.y, q.z};
}
inline f3 inv_project_position(f3 c, const float* T)
{
    f2 n = uv_to_ndc({c.x, c.y});
    f4 p = mul({n.x, n.y, c.z, 1.0f}, T);
    return {p.x / p.w, p.y / p.w, p.z / p.w};
}
inline f3 screen_xy_depth_to_view_space(f3 c, const float* P)
{
    f2 n = uv_to_ndc({c.x, c.y});
    float z = depth_to_camera_z(c.z, P);
    return {z * n.x / P[0], z * n.y / P[5], z};
}
// FEATURE_FLAG_REVERSED_DEPTH (SSAO / SSR / POSTFX _OPTION_INVERTED_DEPTH): every entry point that depends on the depth convention sets this from
// ival[7] before its loops (the checker is called from one thread at a time)
inline bool g_reversed_depth = false;
inline void  set_depth_convention(const ref_args* a) { g_reversed_depth = a->ival[7] != 0; }
inline bool  is_background(float d) { return g_reversed_depth ? d < 1e-6f : d >= (1.0f - 1e-6f); } // SSAO_Common.fxh:16-23 / SSR_Common.fxh:48-55
inline float depth_far_plane() { return g_reversed_depth ? 0.0f : 1.0f; }                         // DepthFarPlane, SSR_Common.fxh:6-12
inline float closest_depth(float a, float b) { return g_reversed_depth ? std::fmax(a, b) : std::fmin(a, b); } // ClosestDepth
inline float luminance601(f3 c) { return dot(c, f3{0.299f, 0.587f, 0.114f}); }            // PostFX_Common.fxh:40
inline float spatial_weight(float d, float sigma) { return std::exp(-d / (2.0f * sigma * sigma)); } // PostFX_Common.fxh:134
inline float bayer4x4(uint32_t px, uint32_t py, uint32_t frame)                            // PostFX_Common.fxh:57-65
{
    uint32_t wx = px & 3u, wy = py & 3u;
    uint32_t A = 2068378560u * (1u - (wx >> 1u)) + 1500172770u * (wx >> 1u);
    uint32_t B = (wy + ((wx & 1u) << 2u)) << 2u;
    return float(((A >> B) + frame) & 0xFu) / 16.0f;
}
inline f2 rotate_vector(f4 r, f2 v) { return {v.x * r.x + v.y * r.y, v.x * r.z + v.y * r.w}; } // PostFX_Common.fxh:80-83
inline float smoothstep(float a, float b, float x) { float t = sat((x - a) / (b - a)); return t * t * (3.0f - 2.0f * t); }

// ------------------------------------------------------------------------------------------------ images
struct Img
{
    const ref_img* im;
    int w() const { return im->w; }
    int h() const { return im->h; }
    const float* px(int x, int y) const { return im->data + (size_t(y) * im->w + x) * im->c; }
    float*       wpx(int x, int y) const { return im->data + (size_t(y) * im->w + x) * im->c; }
    bool  inside(int x, int y) const { return x >= 0 && y >= 0 && x < im->w && y < im->h; }
    float ld1(int x, int y) const { return px(x, y)[0]; }
    f2    ld2(int x, int y) const { const float* p = px(x, y); return {p[0], p[1]}; }
    f4    ld4(int x, int y) const { const float* p = px(x, y); return {p[0], p[1], p[2], p[3]}; }
    f3    ld3(int x, int y) const { const float* p = px(x, y); return {p[0], p[1], p[2]}; }
    float ld1c(int x, int y) const { return ld1(clampi(x, 0, im->w - 1), clampi(y, 0, im->h - 1)); } // clamp addressing
    float ld1z(int x, int y) const { return inside(x, y) ? ld1(x, y) : 0.0f; }                       // D3D Load: out of bounds -> 0
    f2    ld2z(int x, int y) const { return inside(x, y) ? ld2(x, y) : f2{0.f, 0.f}; }
    void  st1(int x, int y, float v) const { wpx(x, y)[0] = v; }
    void  st2(int x, int y, f2 v) const { float* p = wpx(x, y); p[0] = v.x; p[1] = v.y; }
    void  st4(int x, int y, f4 v) const { float* p = wpx(x, y); p[0] = v.x; p[1] = v.y; p[2] = v.z; p[3] = v.w; }
};
inline Img in_img(const ref_args* a, int slot, int mip = 0) { return Img{&a->in[slot][mip]}; }
inline Img out_img(const ref_args* a, int slot) { return Img{&a->out[slot]}; }

struct Bilinear
{
    int x0, y0, x1, y1;
    float w00, w10, w01, w11;
};
inline Bilinear bilinear_uc(float lx, float ly, int w, int h) // GetBilinearSamplingInfoUC
{
    lx -= 0.5f; ly -= 0.5f;
    float fx = std::floor(lx), fy = std::floor(ly);
    Bilinear b;
    b.x0 = clampi(int(fx), 0, w - 1); b.y0 = clampi(int(fy), 0, h - 1);
    b.x1 = clampi(int(fx) + 1, 0, w - 1); b.y1 = clampi(int(fy) + 1, 0, h - 1);
    float x = lx - fx, y = ly - fy;
    b.w00 = (1.0f - x) * (1.0f - y); b.w10 = x * (1.0f - y); b.w01 = (1.0f - x) * y; b.w11 = x * y;
    return b;
}
inline float sample_linear_clamp1(const Img& im, float u, float v)
{
    Bilinear b = bilinear_uc(u * float(im.w()), v * float(im.h()), im.w(), im.h());
    return im.ld1(b.x0, b.y0) * b.w00 + im.ld1(b.x1, b.y0) * b.w10 + im.ld1(b.x0, b.y1) * b.w01 + im.ld1(b.x1, b.y1) * b.w11;
}
inline f4 sample_linear_clamp4(const Img& im, float u, float v)
{
    Bilinear b = bilinear_uc(u * float(im.w()), v * float(im.h()), im.w(), im.h());
    return im.ld4(b.x0, b.y0) * b.w00 + im.ld4(b.x1, b.y0) * b.w10 + im.ld4(b.x0, b.y1) * b.w01 + im.ld4(b.x1, b.y1) * b.w11;
}
inline f2 sample_linear_clamp2(const Img& im, float u, float v)
{
    Bilinear b = bilinear_uc(u * float(im.w()), v * float(im.h()), im.w(), im.h());
    return im.ld2(b.x0, b.y0) * b.w00 + im.ld2(b.x1, b.y0) * b.w10 + im.ld2(b.x0, b.y1) * b.w01 + im.ld2(b.x1, b.y1) * b.w11;
}
inline f3 sample_linear_border3(const Img& im, float u, float v) // linear filter, BORDER addressing with colour 0 (Bloom.cpp:52-59)
{
    float fx = u * float(im.w()) - 0.5f, fy = v * float(im.h()) - 0.5f;
    float x0f = std::floor(fx), y0f = std::floor(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    const float wgt[4] = {(1.0f - wx) * (1.0f - wy), wx * (1.0f - wy), (1.0f - wx) * wy, wx * wy};
    f3 acc{0.f, 0.f, 0.f};
    for (int t = 0; t < 4; ++t)
    {
        int x = x0 + (t & 1), y = y0 + (t >> 1);
        if (im.inside(x, y)) acc += im.ld3(x, y) * wgt[t];
    }
    return acc;
}
inline float sample_point_clamp1(const Img& im, float u, float v)
{
    return im.ld1(clampi(int(std::floor(u * float(im.w()))), 0, im.w() - 1), clampi(int(std::floor(v * float(im.h()))), 0, im.h() - 1));
}

// ------------------------------------------------------------------------------------------------ PBR_Common.fxh
constexpr float kPI = 3.141592653589793f;
inline float dot_sat(f3 a, f3 b) { return sat(dot(a, b)); }
inline float pow5(float x) { float x2 = x * x; return x2 * x2 * x; }
inline f3 schlick_reflection(float VdotH, f3 r0, f3 r90) { return r0 + (r90 - r0) * pow5(clampf(1.0f - VdotH, 0.0f, 1.0f)); } // :81
inline float smith_ggx_visibility_correlated(float NdotL, float NdotV, float alpha) // :107-123
{
    float a2 = alpha * alpha;
    float ggxv = NdotL * std::sqrt(fmax2(NdotV * NdotV * (1.0f - a2) + a2, 1e-7f));
    float ggxl = NdotV * std::sqrt(fmax2(NdotL * NdotL * (1.0f - a2) + a2, 1e-7f));
    return 0.5f / (ggxv + ggxl);
}
inline float smith_ggx_masking(float NdotV, float alpha) // :149-175
{
    float a2 = alpha * alpha;
    float denom = NdotV + std::sqrt(a2 + (1.0f - a2) * NdotV * NdotV);
    return 2.0f * fmax2(NdotV, 0.0f) / fmax2(denom, 1e-6f);
}
inline float normal_distribution_ggx(float NdotH, float alpha) // :181-194
{
    alpha = fmax2(alpha, 1e-3f);
    float a2 = alpha * alpha, nh2 = NdotH * NdotH;
    float f = nh2 * a2 + (1.0f - nh2);
    return a2 / fmax2(kPI * f * f, 1e-9f);
}
inline f3 smith_ggx_sample_visible_normal_sc(f3 view, float ax, float ay, float u1, float u2) // :278-295
{
    f3 V = normalize(view * f3{ax, ay, 1.0f});
    float phi = 2.0f * kPI * u1;
    float z = (1.0f - u2) * (1.0f + V.z) - V.z;
    float st = std::sqrt(clampf(1.0f - z * z, 0.0f, 1.0f));
    f3 H = f3{st * std::cos(phi), st * std::sin(phi), z} + V;
    return normalize(f3{ax * H.x, ay * H.y, H.z});
}
} // namespace ok
