// mifx_oracle.cpp -- TEST INFRASTRUCTURE ONLY: the CPU oracle ("port") of the DiligentFX hot path.
//
// A hand-written, scalar, single-file restatement of the reference's per-pixel algorithms in plain C++ (no HLSL
// shim, no dependence on the product sources).  Every function cites the reference file:line it follows.
// It is pinned against the reference itself: tests/test_oracle_vs_ref.py compares every entry point with
// oracle/_ref/libmifx_ref.so (the reference's shader source compiled for the CPU) and with the golden
// fixtures under tests/golden/ that were generated from it.
//
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only as the
// checker.  The product (diligentfx_amd/libmifx.so) never links, imports or calls it.
//
// Conventions (SURVEY.md Appendix A, D3D/Vulkan path): NDC z in [0,1], UV origin top-left, row-major matrices with
// row-vector multiplication, non-reversed depth, out-of-bounds Load returns 0, pixel centre = (x+0.5, y+0.5).
// Build: g++ -O2 -fopenmp -fsingle-precision-constant -ffp-contract=off (oracle/build.py).
#include <cmath>
#include <cstdint>
#include <cstring>
#include <algorithm>

#include "oracle_args.h"

namespace
{
// ------------------------------------------------------------------------------------------------ tiny vector kit
struct f2 { float x, y; };
struct f3 { float x, y, z; };
struct f4 { float x, y, z, w; };
inline f2 operator+(f2 a, f2 b) { return {a.x + b.x, a.y + b.y}; }
inline f2 operator-(f2 a, f2 b) { return {a.x - b.x, a.y - b.y}; }
inline f2 operator*(f2 a, f2 b) { return {a.x * b.x, a.y * b.y}; }
inline f2 operator*(f2 a, float b) { return {a.x * b, a.y * b}; }
inline f2 operator*(float a, f2 b) { return {a * b.x, a * b.y}; }
inline f3 operator+(f3 a, f3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline f3 operator-(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline f3 operator*(f3 a, f3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline f3 operator/(f3 a, f3 b) { return {a.x / b.x, a.y / b.y, a.z / b.z}; }
inline f3 operator*(f3 a, float b) { return {a.x * b, a.y * b, a.z * b}; }
inline f3 operator*(float a, f3 b) { return {a * b.x, a * b.y, a * b.z}; }
inline f3 operator/(f3 a, float b) { return {a.x / b, a.y / b, a.z / b}; }
inline f3 operator+(f3 a, float b) { return {a.x + b, a.y + b, a.z + b}; }
inline f3 operator+(float a, f3 b) { return {a + b.x, a + b.y, a + b.z}; }
inline f3 operator-(f3 a, float b) { return {a.x - b, a.y - b, a.z - b}; }
inline f3 operator-(f3 a) { return {-a.x, -a.y, -a.z}; }
inline f4 operator+(f4 a, f4 b) { return {a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w}; }
inline f4 operator-(f4 a, f4 b) { return {a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w}; }
inline f4 operator*(f4 a, f4 b) { return {a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w}; }
inline f4 operator*(f4 a, float b) { return {a.x * b, a.y * b, a.z * b, a.w * b}; }
inline f4 operator*(float a, f4 b) { return {a * b.x, a * b.y, a * b.z, a * b.w}; }
inline f4 operator/(f4 a, float b) { return {a.x / b, a.y / b, a.z / b, a.w / b}; }
inline float dot(f2 a, f2 b) { return a.x * b.x + a.y * b.y; }
inline float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline float dot(f4 a, f4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
inline float length(f2 a) { return std::sqrt(dot(a, a)); }
inline float length(f3 a) { return std::sqrt(dot(a, a)); }
inline f3 normalize(f3 a) { return a * (1.0f / std::sqrt(dot(a, a))); }
inline f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline f3 reflect(f3 i, f3 n) { return i - 2.0f * dot(n, i) * n; }
// D3D / IEEE-754-2008 minNum/maxNum semantics: a NaN operand is ignored; saturate(NaN) = 0
inline float fmin2(float a, float b) { return std::fmin(a, b); }
inline float fmax2(float a, float b) { return std::fmax(a, b); }
inline float sat(float x) { return fmin2(fmax2(x, 0.0f), 1.0f); }
inline float clampf(float x, float a, float b) { return fmin2(fmax2(x, a), b); }
inline int   clampi(int x, int a, int b) { return x < a ? a : (x > b ? b : x); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline f3 lerp(f3 a, f3 b, float t) { return a + t * (b - a); }
inline f3 lerp(f3 a, f3 b, f3 t) { return a + t * (b - a); }
inline f4 lerp(f4 a, f4 b, float t) { return a + t * (b - a); }
inline float frac(float x) { return x - std::floor(x); }
inline f3 max3(f3 a, float b) { return {fmax2(a.x, b), fmax2(a.y, b), fmax2(a.z, b)}; }
inline f3 pow3(f3 a, float e) { return {std::pow(a.x, e), std::pow(a.y, e), std::pow(a.z, e)}; }
inline f3 splat(float s) { return {s, s, s}; }
inline f3 xyz(f4 a) { return {a.x, a.y, a.z}; }

// ------------------------------------------------------------------------------------------------ camera (CameraAttribs, BasicStructures.fxh:84-149)
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

inline f4 mul(f4 v, const float* M) // row vector x row-major matrix
{
    return {v.x * M[0] + v.y * M[4] + v.z * M[8] + v.w * M[12], v.x * M[1] + v.y * M[5] + v.z * M[9] + v.w * M[13],
            v.x * M[2] + v.y * M[6] + v.z * M[10] + v.w * M[14], v.x * M[3] + v.y * M[7] + v.z * M[11] + v.w * M[15]};
}
// ShaderUtilities.fxh:5-40
inline float camera_z_to_depth(float z, const float* P) { return (P[10] * z + P[14]) / (P[11] * z + P[15]); }
inline float depth_to_camera_z(float d, const float* P) { return (P[14] - d * P[15]) / (d * P[11] - P[10]); }
// Appendix A
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

// ------------------------------------------------------------------------------------------------ image access
inline const float* texel(const ref_img& im, int x, int y) { return im.data + (size_t(y) * im.w + x) * im.c; }
inline float* texel_w(const ref_img& im, int x, int y) { return im.data + (size_t(y) * im.w + x) * im.c; }
inline float load1_zero(const ref_img& im, int x, int y) { return (x < 0 || y < 0 || x >= im.w || y >= im.h) ? 0.0f : texel(im, x, y)[0]; }
inline f2 load2_zero(const ref_img& im, int x, int y)
{
    if (x < 0 || y < 0 || x >= im.w || y >= im.h) return {0.f, 0.f};
    const float* p = texel(im, x, y);
    return {p[0], p[1]};
}
inline f4 load4(const ref_img& im, int x, int y)
{
    const float* p = texel(im, x, y);
    return {p[0], p[1], p[2], p[3]};
}

// ------------------------------------------------------------------------------------------------ sRGB (SRGBUtilities.fxh:4-33)
inline float srgb_to_linear1(float s)
{
    float less = s >= 0.04045f ? 1.0f : 0.0f;
    return lerp(s / 12.92f, std::pow(sat((s + 0.055f) / 1.055f), 2.4f), less);
}
inline float linear_to_srgb1(float s)
{
    float gr = s >= 0.0031308f ? 1.0f : 0.0f;
    return lerp(s * 12.92f, std::pow(s, 1.0f / 2.4f) * 1.055f - 0.055f, gr);
}

// ------------------------------------------------------------------------------------------------ ToneMap (ToneMapping.fxh:8-226)
struct ToneMappingAttribs // ToneMappingStructures.fxh:24-52
{
    int32_t mode, autoExposure; float middleGray; int32_t lightAdaptation;
    float whitePoint, lumSaturation; uint32_t pad0, pad1;
    float agxSaturation, agxSlope, agxPower, agxOffset;
};
static_assert(sizeof(ToneMappingAttribs) == 48, "ToneMappingAttribs layout");

inline f3 uncharted2(f3 x) // :8-19
{
    const float A = 0.15f, B = 0.50f, C = 0.10f, D = 0.20f, E = 0.02f, F = 0.30f;
    return ((x * (A * x + C * B) + D * E) / (x * (A * x + B) + D * F)) - E / F;
}
inline f3 agx_contrast(f3 x) // :21-33
{
    f3 x2 = x * x, x4 = x2 * x2;
    return 15.5f * x4 * x2 - 40.14f * x4 * x + 31.96f * x4 - 6.868f * x2 * x + 0.4298f * x2 + 0.1191f * x - 0.00232f;
}
inline f3 agx(f3 c) // :35-56
{
    const f3 r0{0.842479062253094f, 0.0784335999999992f, 0.0792237451477643f};
    const f3 r1{0.0423282422610123f, 0.878468636469772f, 0.0791661274605434f};
    const f3 r2{0.0423756549057051f, 0.0784336f, 0.879142973793104f};
    const float MinEv = -12.47393f, MaxEv = 4.026069f;
    c = {dot(r0, c), dot(r1, c), dot(r2, c)};
    c = {clampf(std::log2(c.x), MinEv, MaxEv), clampf(std::log2(c.y), MinEv, MaxEv), clampf(std::log2(c.z), MinEv, MaxEv)};
    c = (c - MinEv) / (MaxEv - MinEv);
    return agx_contrast(c);
}
inline f3 agx_eotf(f3 c) // :58-72
{
    const f3 r0{+1.19687900512017f, -0.0980208811401368f, -0.0990297440797205f};
    const f3 r1{-0.0528968517574562f, +1.15190312990417f, -0.0989611768448433f};
    const f3 r2{-0.0529716355144438f, -0.0980434501171241f, +1.15107367264116f};
    c = {dot(r0, c), dot(r1, c), dot(r2, c)};
    return {srgb_to_linear1(c.x), srgb_to_linear1(c.y), srgb_to_linear1(c.z)};
}
inline f3 agx_look(f3 c, float sat_, float offset, float slope, float power) // :74-85
{
    float lum = dot(c, f3{0.212671f, 0.715160f, 0.072169f});
    c = pow3(c * slope + offset, power);
    return lum + sat_ * (c - lum);
}

f3 tone_map(f3 color, const ToneMappingAttribs& a, float aveLogLum) // :87-226
{
    const f3 W{0.212671f, 0.715160f, 0.072169f};
    float lumScale = a.middleGray / aveLogLum;
    color = max3(color, 0.0f);
    float pixLum = fmax2(dot(W, color), 1e-10f);
    float scaledLum = pixLum * lumScale;
    f3 scaled = color * lumScale;
    float wp = a.whitePoint;
    switch (a.mode)
    {
        case 1: { float t = 1.0f - std::exp(-scaledLum); return t * pow3(color / pixLum, a.lumSaturation); }
        case 2: { float t = scaledLum / (1.0f + scaledLum); return t * pow3(color / pixLum, a.lumSaturation); }
        case 3: { float t = scaledLum * (1.0f + scaledLum / (wp * wp)) / (1.0f + scaledLum); return t * pow3(color / pixLum, a.lumSaturation); }
        case 4: { f3 curr = uncharted2(2.0f * scaled); f3 white = splat(1.0f) / uncharted2(splat(wp)); return curr * white; }
        case 5:
        {
            f3 t = max3(scaled - splat(0.004f), 0.0f);
            t = (t * (6.2f * t + splat(0.5f))) / (t * (6.2f * t + splat(1.7f)) + splat(0.06f));
            return pow3(t, 2.2f);
        }
        case 6: { float t = std::log10(1.0f + scaledLum) / std::log10(1.0f + wp); return t * pow3(color / pixLum, a.lumSaturation); }
        case 7:
        {
            const float Bias = 0.85f;
            float t = 1.0f / std::log10(1.0f + wp) * std::log(1.0f + scaledLum) / std::log(2.0f + 8.0f * std::pow(scaledLum / wp, std::log(Bias) / std::log(0.5f)));
            return t * pow3(color / pixLum, a.lumSaturation);
        }
        case 8: return agx_eotf(agx(scaled));
        case 9: return agx_eotf(agx_look(agx(scaled), a.agxSaturation, a.agxOffset, a.agxSlope, a.agxPower));
        case 10:
        {
            color = color * (0.3f / aveLogLum);
            const float StartCompression = 0.8f - 0.04f, Desaturation = 0.15f;
            float x = fmin2(color.x, fmin2(color.y, color.z));
            float offset = x < 0.08f ? x - 6.25f * x * x : 0.04f;
            color = color - offset;
            float peak = fmax2(color.x, fmax2(color.y, color.z));
            if (peak >= StartCompression)
            {
                float d = 1.0f - StartCompression;
                float newPeak = 1.0f - d * d / (peak + d - StartCompression);
                color = color * (newPeak / peak);
                float g = 1.0f - 1.0f / (Desaturation * (peak - newPeak) + 1.0f);
                color = lerp(color, splat(newPeak), g);
            }
            return color;
        }
        case 11:
        {
            color = color * (0.3f / aveLogLum);
            const float StartCompression = 0.8f, Desaturation = 0.5f;
            float d = 1.0f - StartCompression;
            float peak = fmax2(color.x, fmax2(color.y, color.z));
            if (peak >= StartCompression)
            {
                float newPeak = 1.0f - d * d / (peak + d - StartCompression);
                float invPeak = 1.0f / peak;
                float extra = dot(color * (1.0f - StartCompression * invPeak), splat(1.0f));
                color = color * (newPeak * invPeak);
                float g = 1.0f - 3.0f / (Desaturation * extra + 3.0f);
                color = lerp(color, splat(1.0f), g);
            }
            return color;
        }
        default: return color;
    }
}
} // namespace

extern "C" {

// M2 -- full-screen ToneMap as in Hydrogent/shaders/HnCopyFrame.psh:27-36,61-63.
// in[0]: HDR colour (c=4); out[0]: colour (c=4); attribs: ToneMappingAttribs; fval[0]: fAveLogLum; ival[0]: 1 = LinearToSRGB
int oracle_tonemap(const ref_args* a)
{
    ToneMappingAttribs attr;
    std::memcpy(&attr, a->attribs, sizeof(attr));
    if (attr.mode < 0 || attr.mode > 11) return -1;
    const ref_img& in = a->in[0][0];
    const ref_img& out = a->out[0];
    const float lum = a->fval[0];
    const bool srgb = a->ival[0] != 0;
#pragma omp parallel for
    for (int y = 0; y < in.h; ++y)
        for (int x = 0; x < in.w; ++x)
        {
            f4 c = load4(in, x, y);
            f3 t = tone_map(xyz(c), attr, lum);
            if (srgb) t = {linear_to_srgb1(t.x), linear_to_srgb1(t.y), linear_to_srgb1(t.z)};
            float* o = texel_w(out, x, y);
            o[0] = t.x; o[1] = t.y; o[2] = t.z; o[3] = c.w;
        }
    return 0;
}

// C1 -- Shaders/Common/private/ComputeBlueNoiseTexture.fx:18-89; targets are RG8_UNORM (PostFXContext.cpp:200).
// in[0]: Sobol 256x1 (c=1, byte values as floats), in[1]: scrambling tile 512x256; out[0]: XY (128x128, c=2), out[1]: ZW; ival[0]: frame index
int oracle_blue_noise(const ref_args* a)
{
    const ref_img& sobol = a->in[0][0];
    const ref_img& tile = a->in[1][0];
    const uint32_t frame = uint32_t(a->ival[0]);
    auto sample = [&](uint32_t px, uint32_t py, uint32_t dim) { // SampleRandomNumber :18-32
        px &= 127u; py &= 127u; dim &= 255u;
        uint32_t value = uint32_t(sobol.data[dim]);
        uint32_t idx = (dim % 8u) + (px + py * 128u) * 8u;
        value ^= uint32_t(tile.data[(idx / 512u) * 512u + (idx % 512u)]);
        return (float(value) + 0.5f) / 256.0f;
    };
    auto hilbert = [](uint32_t px, uint32_t py) { // HilbertIndex :34-57
        const uint32_t W = 128u;
        px &= W - 1u; py &= W - 1u;
        uint32_t index = 0u;
        for (uint32_t lvl = W / 2u; lvl > 0u; lvl /= 2u)
        {
            uint32_t rx = (px & lvl) > 0u, ry = (py & lvl) > 0u;
            index += lvl * lvl * ((3u * rx) ^ ry);
            if (ry == 0u)
            {
                if (rx == 1u) { px = (W - 1u) - px; py = (W - 1u) - py; }
                std::swap(px, py);
            }
        }
        return index;
    };
    auto unorm8 = [](float v) { v = sat(v); return std::floor(v * 255.0f + 0.5f) / 255.0f; };
    for (uint32_t y = 0; y < 128u; ++y)
        for (uint32_t x = 0; x < 128u; ++x)
        {
            const float G = 1.61803398875f; // SampleRandomVector2D :60-68
            float alpha = 0.5f + (1.0f / G) * float(frame & 0xFFu);
            float* oxy = texel_w(a->out[0], int(x), int(y));
            oxy[0] = unorm8(frac(sample(x, y, 0u) + alpha));
            oxy[1] = unorm8(frac(sample(x, y, 1u) + alpha));
            uint32_t index = hilbert(x, y) + frame; // SampleRandomVector1D1D :71-79
            index += 288u * (frame & 127u);
            const float G2 = 1.32471795724474602596f;
            float* ozw = texel_w(a->out[1], int(x), int(y));
            ozw[0] = unorm8(frac(0.5f + float(index) * (1.0f / G2)));
            ozw[1] = unorm8(frac(0.5f + float(index) * (1.0f / (G2 * G2))));
        }
    return 0;
}

// C2 -- Shaders/Common/private/ComputeReprojectedDepth.fx:18-30. in[0]: depth; cam0, cam1; out[0]: reprojected depth
int oracle_reprojected_depth(const ref_args* a)
{
    Camera cur, prev;
    std::memcpy(&cur, a->cam0, sizeof(Camera));
    std::memcpy(&prev, a->cam1, sizeof(Camera));
    const ref_img& depth = a->in[0][0];
    const ref_img& out = a->out[0];
#pragma omp parallel for
    for (int y = 0; y < out.h; ++y)
        for (int x = 0; x < out.w; ++x)
        {
            float d = texel(depth, x, y)[0];
            f3 sc{(float(x) + 0.5f) * cur.viewport[2], (float(y) + 0.5f) * cur.viewport[3], d};
            sc.x += 0.5f * cur.jitter[0];
            sc.y += -0.5f * cur.jitter[1];
            f3 world = inv_project_position(sc, cur.viewProjInv);
            f3 p = project_position(world, prev.viewProj);
            texel_w(out, x, y)[0] = p.z;
        }
    return 0;
}

// C3 -- Shaders/Common/private/ComputeClosestMotion.fx:24-55 (3x3 search, x outer / y inner, strict '<', unclamped loads -> 0)
// in[0]: depth, in[1]: motion (c=2); out[0]: closest motion (c=2)
int oracle_closest_motion(const ref_args* a)
{
    const ref_img& depth = a->in[0][0];
    const ref_img& motion = a->in[1][0];
    const ref_img& out = a->out[0];
    const bool reversed = a->ival[7] != 0; // POSTFX_OPTION_INVERTED_DEPTH
#pragma omp parallel for
    for (int y = 0; y < out.h; ++y)
        for (int x = 0; x < out.w; ++x)
        {
            float closest = reversed ? 0.0f : 1.0f; // DepthFarPlane, ComputeClosestMotion.fx:5-9
            int ox = 0, oy = 0;
            for (int dx = -1; dx <= 1; ++dx)
                for (int dy = -1; dy <= 1; ++dy)
                {
                    float nd = load1_zero(depth, x + dx, y + dy);
                    if (reversed ? nd > closest : nd < closest) { ox = dx; oy = dy; closest = nd; } // :36-40
                }
            f2 m = load2_zero(motion, x + ox, y + oy);
            float* o = texel_w(out, x, y);
            o[0] = m.x; o[1] = m.y;
        }
    return 0;
}

} // extern "C"
