// mifx_device.h -- device-side building blocks shared by all HIP kernels of libmifx:
// pitched image views, small fp32 vector types, the DiligentCore D3D/Vulkan conventions
// and the shared shader-library functions (depth<->camera Z, projection helpers, software texture filtering).
//
// Software filtering contract: exact fp32 bilinear weights as spelled out by GetBilinearSamplingInfoUC
// (Shaders/Common/public/ShaderUtilities.fxh:126-142); point-mip selection = floor(lod + 0.5).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mifx
{
#define MIFX_HD __host__ __device__ __forceinline__
#define MIFX_D __device__ __forceinline__
// minimum waves per SIMD the register allocator must leave room for (caps VGPRs at 512 / n): straight-line filter kernels otherwise hoist
// all their taps into registers and drop to 2-3 waves per SIMD, too few to hide the load latency
#define MIFX_WAVES(n) __attribute__((amdgpu_waves_per_eu(n)))
#define MIFX_WAVES_OPT_0
#define MIFX_WAVES_OPT_3 MIFX_WAVES(3)
#define MIFX_WAVES_OPT_4 MIFX_WAVES(4)
#define MIFX_WAVES_OPT_5 MIFX_WAVES(5)
#define MIFX_WAVES_OPT_6 MIFX_WAVES(6)
#define MIFX_WAVES_OPT_7 MIFX_WAVES(7)
#define MIFX_WAVES_OPT_8 MIFX_WAVES(8)
#define MIFX_WAVES_CAT(n) MIFX_WAVES_OPT_##n
#define MIFX_WAVES_OPT(n) MIFX_WAVES_CAT(n) // experiment knob: MIFX_WAVES_OPT(0) = no hint
// Weighted sums of fetched texels (filter taps) may fuse each multiply-add: one rounding instead of two, half the instructions.  Only for
// smooth accumulations whose result does not steer addressing or thresholds; everything else keeps the reference's separate mul / add.
#define MIFX_FMA_BLOCK _Pragma("clang fp contract(fast)")
// scheduling fence between filter taps: the tap result must be complete here and no memory access moves across, so the scheduler cannot
// hoist every tile read of a 9- / 13-tap filter above the arithmetic (which cost 195-256 VGPRs and left 1-2 waves per SIMD)
#define MIFX_TAP_FENCE(v) __asm__ volatile("" : "+v"((v).x), "+v"((v).y), "+v"((v).z) : : "memory")

// ------------------------------------------------------------------------------------------------ vectors
struct v2 { float x, y; };
struct v3 { float x, y, z; };
struct v4 { float x, y, z, w; };
struct i2 { int x, y; };

MIFX_HD v2 mk2(float x, float y) { return v2{x, y}; }
MIFX_HD v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
MIFX_HD v3 mk3(float s) { return v3{s, s, s}; }
MIFX_HD v4 mk4(float x, float y, float z, float w) { return v4{x, y, z, w}; }
MIFX_HD v4 mk4(v3 a, float w) { return v4{a.x, a.y, a.z, w}; }
MIFX_HD v4 mk4(float s) { return v4{s, s, s, s}; }
MIFX_HD v3 xyz(v4 a) { return v3{a.x, a.y, a.z}; }

// ------------------------------------------------------------------------------------------------ division
// fdiv(a, b): a / b in 5 instructions instead of the ~11 of the compiler's correctly rounded expansion (the chain is VALU-issue bound and
// divisions are 20-40 % of it): q = a * rcp(b), one FMA residual correction, v_div_fixup for zero / infinite / NaN operands (the TAA colour
// clip divides by zero on purpose).  The corrected quotient differs from the IEEE one only when the exact quotient lies within ~2^-23 ulp
// of a rounding boundary (about one quotient in four million, by 1 ulp), so it is safe in front of thresholds and texel selection.  Unlike
// the compiler's expansion it does not pre-scale: a denormal divisor or a quotient that overflows / underflows is not handled (no such
// operands on this path: divisors are view-space depths, texture sizes, weights clamped away from zero).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIFX_PRECISE_MATH)
MIFX_HD float fdiv(float a, float b)
{
    if (__builtin_constant_p(b) && (__builtin_bit_cast(unsigned, b) & 0x007fffffu) == 0u && __builtin_fabsf(b) > 1e-30f && __builtin_fabsf(b) < 1e30f)
        return a * (1.0f / b); // power-of-two divisor known at compile time (after inlining): the reciprocal multiply is exact
    const float r = __builtin_constant_p(b) ? 1.0f / b : __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    const float e = __builtin_fmaf(-q, b, a);
    return __builtin_amdgcn_div_fixupf(__builtin_fmaf(e, r, q), b, a);
}
#else
MIFX_HD float fdiv(float a, float b) { return a / b; }
#endif

// fsqrt(x): correctly rounded sqrt for normal x (and 0, inf, NaN) -- the hardware 1-ulp estimate followed by the next-down / next-up residual
// test of the compiler's IEEE expansion, without that expansion's rescaling of denormal inputs and operand classification (13 issue slots
// instead of 19).  Bit-identical to sqrtf() on everything this path feeds it (squared lengths, variances, roughness).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIFX_PRECISE_MATH)
MIFX_HD float fsqrt(float x)
{
    const float s  = __builtin_amdgcn_sqrtf(x);
    const float sd = __builtin_bit_cast(float, __builtin_bit_cast(int, s) - 1);
    const float su = __builtin_bit_cast(float, __builtin_bit_cast(int, s) + 1);
    const float rd = __builtin_fmaf(-sd, s, x);
    const float ru = __builtin_fmaf(-su, s, x);
    float r = rd <= 0.0f ? sd : s;
    r = ru > 0.0f ? su : r;
    return r;
}
#else
MIFX_HD float fsqrt(float x) { return sqrtf(x); }
#endif

#define MIFX_VEC_OPS2(op)                                                   \
    MIFX_HD v2 operator op(v2 a, v2 b) { return v2{a.x op b.x, a.y op b.y}; } \
    MIFX_HD v2 operator op(v2 a, float b) { return v2{a.x op b, a.y op b}; }  \
    MIFX_HD v2 operator op(float a, v2 b) { return v2{a op b.x, a op b.y}; }
#define MIFX_VEC_OPS3(op)                                                               \
    MIFX_HD v3 operator op(v3 a, v3 b) { return v3{a.x op b.x, a.y op b.y, a.z op b.z}; } \
    MIFX_HD v3 operator op(v3 a, float b) { return v3{a.x op b, a.y op b, a.z op b}; }    \
    MIFX_HD v3 operator op(float a, v3 b) { return v3{a op b.x, a op b.y, a op b.z}; }
#define MIFX_VEC_OPS4(op)                                                                           \
    MIFX_HD v4 operator op(v4 a, v4 b) { return v4{a.x op b.x, a.y op b.y, a.z op b.z, a.w op b.w}; } \
    MIFX_HD v4 operator op(v4 a, float b) { return v4{a.x op b, a.y op b, a.z op b, a.w op b}; }      \
    MIFX_HD v4 operator op(float a, v4 b) { return v4{a op b.x, a op b.y, a op b.z, a op b.w}; }
MIFX_VEC_OPS2(+) MIFX_VEC_OPS2(-) MIFX_VEC_OPS2(*)
MIFX_VEC_OPS3(+) MIFX_VEC_OPS3(-) MIFX_VEC_OPS3(*)
MIFX_VEC_OPS4(+) MIFX_VEC_OPS4(-) MIFX_VEC_OPS4(*)
MIFX_HD v2 operator/(v2 a, v2 b) { return v2{fdiv(a.x, b.x), fdiv(a.y, b.y)}; }
MIFX_HD v2 operator/(v2 a, float b) { return v2{fdiv(a.x, b), fdiv(a.y, b)}; }
MIFX_HD v2 operator/(float a, v2 b) { return v2{fdiv(a, b.x), fdiv(a, b.y)}; }
MIFX_HD v3 operator/(v3 a, v3 b) { return v3{fdiv(a.x, b.x), fdiv(a.y, b.y), fdiv(a.z, b.z)}; }
MIFX_HD v3 operator/(v3 a, float b) { return v3{fdiv(a.x, b), fdiv(a.y, b), fdiv(a.z, b)}; }
MIFX_HD v3 operator/(float a, v3 b) { return v3{fdiv(a, b.x), fdiv(a, b.y), fdiv(a, b.z)}; }
MIFX_HD v4 operator/(v4 a, v4 b) { return v4{fdiv(a.x, b.x), fdiv(a.y, b.y), fdiv(a.z, b.z), fdiv(a.w, b.w)}; }
MIFX_HD v4 operator/(v4 a, float b) { return v4{fdiv(a.x, b), fdiv(a.y, b), fdiv(a.z, b), fdiv(a.w, b)}; }
MIFX_HD v4 operator/(float a, v4 b) { return v4{fdiv(a, b.x), fdiv(a, b.y), fdiv(a, b.z), fdiv(a, b.w)}; }
MIFX_HD v2 operator-(v2 a) { return v2{-a.x, -a.y}; }
MIFX_HD v3 operator-(v3 a) { return v3{-a.x, -a.y, -a.z}; }
MIFX_HD v3& operator+=(v3& a, v3 b) { a = a + b; return a; }
MIFX_HD v4& operator+=(v4& a, v4 b) { a = a + b; return a; }

// ------------------------------------------------------------------------------------------------ transcendental math
// Device code uses the CDNA hardware transcendentals (v_exp_f32 / v_log_f32 / v_sin_f32 / v_cos_f32, ~1e-6 relative error), three
// orders of magnitude inside the 1e-3 parity contract; -DMIFX_PRECISE_MATH switches to the correctly rounded libm versions.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIFX_PRECISE_MATH)
MIFX_HD float m_exp(float x) { return __expf(x); }
MIFX_HD float m_exp2(float x) { return __exp2f(x); }
MIFX_HD float m_log(float x) { return __logf(x); }
MIFX_HD float m_log2(float x) { return log2f(x); } // feeds floor(): keep the libm version
MIFX_HD float m_log10(float x) { return __log10f(x); }
MIFX_HD float m_pow(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); } // x >= 0 in every use; __powf is the slow full-precision pow
MIFX_HD float m_sin(float x) { return sinf(x); }   // directions / rotations select texels: keep the libm versions
MIFX_HD float m_cos(float x) { return cosf(x); }
#else
MIFX_HD float m_exp(float x) { return expf(x); }
MIFX_HD float m_exp2(float x) { return exp2f(x); }
MIFX_HD float m_log(float x) { return logf(x); }
MIFX_HD float m_log2(float x) { return log2f(x); }
MIFX_HD float m_log10(float x) { return log10f(x); }
MIFX_HD float m_pow(float x, float y) { return powf(x, y); }
MIFX_HD float m_sin(float x) { return sinf(x); }
MIFX_HD float m_cos(float x) { return cosf(x); }
#endif

// "quick" hardware approximations (1 ulp rcp / sqrt / rsq, v_sin / v_cos): ONLY for smooth, well-conditioned expressions whose results do
// not steer addressing, thresholds or cancelling differences (each use site says why it qualifies)
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIFX_PRECISE_MATH)
MIFX_HD float q_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
MIFX_HD float q_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
MIFX_HD float q_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
MIFX_HD float q_sin(float x) { return __sinf(x); }
MIFX_HD float q_cos(float x) { return __cosf(x); }
#else
MIFX_HD float q_rcp(float x) { return 1.0f / x; }
MIFX_HD float q_sqrt(float x) { return sqrtf(x); }
MIFX_HD float q_rsqrt(float x) { return 1.0f / sqrtf(x); }
MIFX_HD float q_sin(float x) { return sinf(x); }
MIFX_HD float q_cos(float x) { return cosf(x); }
#endif

// sin and cos of an argument of a few turns at most (|x| < ~13; every use is a bounded angle): two-constant Cody-Waite reduction by pi/2
// with FMA, Cephes degree-7 / degree-8 polynomials on [-pi/4, pi/4].  Measured against double precision over [-13, 13]: <= 1.55 ulp,
// 9.3e-8 absolute -- the accuracy class of libm's sinf / cosf (which are not bit-identical to the CPU reference's either) at half their
// instruction count (no large-argument path, both results from one reduction).
MIFX_HD void m_sincos(float x, float& sn, float& cs)
{
    const float j = __builtin_rintf(x * 0.63661977236758134f);
    float r = __builtin_fmaf(-j, 1.57079637050628662109375f, x);
    r = __builtin_fmaf(-j, -4.37113900018624283e-8f, r);
    const float r2 = r * r;
    float ps = __builtin_fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
    ps = __builtin_fmaf(ps, r2, -1.6666654611e-1f);
    const float s = __builtin_fmaf(ps * r2, r, r);
    float pc = __builtin_fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
    pc = __builtin_fmaf(pc, r2, 4.166664568298827e-2f);
    const float c = __builtin_fmaf(pc * r2, r2, __builtin_fmaf(-0.5f, r2, 1.0f));
    const int   q = int(j);
    const float a = (q & 1) ? c : s, b = (q & 1) ? s : c;
    sn = (q & 2) ? -a : a;
    cs = ((q + 1) & 2) ? -b : b;
}
MIFX_HD float m_cos_bounded(float x) { float s, c; m_sincos(x, s, c); return c; }
MIFX_HD float m_sin_bounded(float x) { float s, c; m_sincos(x, s, c); return s; }

MIFX_HD float saturate(float x) { return fminf(fmaxf(x, 0.0f), 1.0f); }
MIFX_HD float lerpf(float a, float b, float t) { return a + t * (b - a); }
MIFX_HD float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
MIFX_HD float rcpf(float x) { return fdiv(1.0f, x); }
MIFX_HD float fracf(float x) { return x - floorf(x); }
MIFX_HD float signf(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
MIFX_HD int   clampi(int x, int a, int b) { return min(max(x, a), b); } // a <= b at every call site: v_max_i32 + v_min_i32 (one v_med3_i32 instead: measured, no gain, A7 +11 %)
MIFX_HD float dot(v2 a, v2 b) { return a.x * b.x + a.y * b.y; }
MIFX_HD float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MIFX_HD float dot(v4 a, v4 b) { return a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w; }
MIFX_HD float length(v2 a) { return fsqrt(dot(a, a)); }
MIFX_HD float length(v3 a) { return fsqrt(dot(a, a)); }
MIFX_HD v3    normalize(v3 a) { return a * fdiv(1.0f, fsqrt(dot(a, a))); }
MIFX_HD v3    cross(v3 a, v3 b) { return v3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
MIFX_HD v3    reflect(v3 i, v3 n) { return i - 2.0f * dot(n, i) * n; }
MIFX_HD v3    lerp3(v3 a, v3 b, float t) { return a + t * (b - a); }
MIFX_HD v3    lerp3(v3 a, v3 b, v3 t) { return a + t * (b - a); }
MIFX_HD v4    lerp4(v4 a, v4 b, float t) { return a + t * (b - a); }
MIFX_HD v3    max3(v3 a, v3 b) { return v3{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z)}; }
MIFX_HD v3    min3(v3 a, v3 b) { return v3{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z)}; }
MIFX_HD v3    max3(v3 a, float b) { return v3{fmaxf(a.x, b), fmaxf(a.y, b), fmaxf(a.z, b)}; }
MIFX_HD v4    max4(v4 a, float b) { return v4{fmaxf(a.x, b), fmaxf(a.y, b), fmaxf(a.z, b), fmaxf(a.w, b)}; }
MIFX_HD v4    max4(v4 a, v4 b) { return v4{fmaxf(a.x, b.x), fmaxf(a.y, b.y), fmaxf(a.z, b.z), fmaxf(a.w, b.w)}; }
MIFX_HD v4    min4(v4 a, v4 b) { return v4{fminf(a.x, b.x), fminf(a.y, b.y), fminf(a.z, b.z), fminf(a.w, b.w)}; }
MIFX_HD v4    sqrt4(v4 a) { return v4{fsqrt(a.x), fsqrt(a.y), fsqrt(a.z), fsqrt(a.w)}; }
MIFX_HD v3    sqrt3(v3 a) { return v3{fsqrt(a.x), fsqrt(a.y), fsqrt(a.z)}; }
MIFX_HD v3    pow3(v3 a, float e) { return v3{m_pow(a.x, e), m_pow(a.y, e), m_pow(a.z, e)}; }
MIFX_HD float max_comp(v3 a) { return fmaxf(a.x, fmaxf(a.y, a.z)); }
MIFX_HD float min_comp(v3 a) { return fminf(a.x, fminf(a.y, a.z)); }

// row-major 4x4, row-vector convention: mul(v, M) = v.x*row0 + v.y*row1 + v.z*row2 + v.w*row3
struct m44 { float m[16]; };
MIFX_HD v4 mul(v4 v, const m44& M)
{
    return v4{v.x * M.m[0] + v.y * M.m[4] + v.z * M.m[8] + v.w * M.m[12],
              v.x * M.m[1] + v.y * M.m[5] + v.z * M.m[9] + v.w * M.m[13],
              v.x * M.m[2] + v.y * M.m[6] + v.z * M.m[10] + v.w * M.m[14],
              v.x * M.m[3] + v.y * M.m[7] + v.z * M.m[11] + v.w * M.m[15]};
}
// direction transform: mul(float4(d, 0), M).xyz
MIFX_HD v3 mul_dir(v3 d, const m44& M)
{
    return v3{d.x * M.m[0] + d.y * M.m[4] + d.z * M.m[8], d.x * M.m[1] + d.y * M.m[5] + d.z * M.m[9], d.x * M.m[2] + d.y * M.m[6] + d.z * M.m[10]};
}

// ------------------------------------------------------------------------------------------------ camera subset passed by value to kernels
struct CamK
{
    m44   view, proj, viewProj, viewInv, viewProjInv;
    float pos[3];
    float vw, vh, ivw, ivh; // f4ViewportSize
    float jx, jy;           // f2Jitter
    uint32_t frameIndex;
    int      reversedDepth; // PostFXContext::FEATURE_FLAG_REVERSED_DEPTH: near plane = depth 1, far plane / background = depth 0
};

// ------------------------------------------------------------------------------------------------ D3D/Vulkan conventions
MIFX_HD v2 ndc_to_uv(v2 xy) { return v2{0.5f + 0.5f * xy.x, 0.5f - 0.5f * xy.y}; }   // NormalizedDeviceXYToTexUV
MIFX_HD v2 uv_to_ndc(v2 uv) { return v2{(uv.x - 0.5f) * 2.0f, (uv.y - 0.5f) * -2.0f}; } // TexUVToNormalizedDeviceXY
// F3NDC_XYZ_TO_UVD_SCALE = (0.5, -0.5, 1)

// Shaders/Common/public/ShaderUtilities.fxh:5-40
MIFX_HD float camera_z_to_depth(float z, const m44& P) { return fdiv(P.m[10] * z + P.m[14], P.m[11] * z + P.m[15]); }
MIFX_HD float depth_to_camera_z(float d, const m44& P) { return fdiv(P.m[14] - d * P.m[15], d * P.m[11] - P.m[10]); }

// Shaders/Common/public/PostFX_Common.fxh:85-111
MIFX_HD v3 project_position(v3 o, const m44& T)
{
    v4 p = mul(mk4(o, 1.0f), T);
    v3 q = xyz(p) / p.w;
    v2 uv = ndc_to_uv(mk2(q.x, q.y));
    return v3{uv.x, uv.y, q.z};
}
MIFX_HD v3 inv_project_position(v3 c, const m44& T)
{
    v2 n = uv_to_ndc(mk2(c.x, c.y));
    v4 p = mul(mk4(n.x, n.y, c.z, 1.0f), T);
    return xyz(p) / p.w;
}
// view-space position from screen uv and CAMERA-space z (the second half of ScreenXYDepthToViewSpace); the SSAO passes read z from the
// camera-z pyramid that A2 writes beside the depth pyramid instead of converting every tap again
// fdiv(a, b) for a finite numerator and a finite, normal, non-zero divisor: the quotient of fdiv() without v_div_fixup_f32 -- that instruction only substitutes the
// special results (zero / infinite / NaN operands, which these operands exclude) and costs two issue slots of the seven.  Like fdiv() it is the correctly rounded
// quotient except when the exact quotient lies within ~2^-23 ulp of a rounding boundary (about one in four million, by one ulp), and a denormal intermediate is not
// handled.  Used where the divisor is a projection scale (uniform per launch) and the numerator a view-space length.
#if defined(__HIP_DEVICE_COMPILE__) && !defined(MIFX_PRECISE_MATH)
MIFX_HD float fdiv_finite(float a, float b)
{
    const float r = __builtin_amdgcn_rcpf(b);
    const float q = a * r;
    return __builtin_fmaf(__builtin_fmaf(-q, b, a), r, q);
}
#else
MIFX_HD float fdiv_finite(float a, float b) { return a / b; }
#endif
MIFX_HD v3 screen_xy_camz_to_view_space(float u, float v, float z, const m44& P)
{
    const v2 n = uv_to_ndc(mk2(u, v));
    return v3{fdiv_finite(z * n.x, P.m[0]), fdiv_finite(z * n.y, P.m[5]), z};
}
MIFX_HD v3 screen_xy_depth_to_view_space(v3 c, const m44& P) { return screen_xy_camz_to_view_space(c.x, c.y, depth_to_camera_z(c.z, P), P); }
MIFX_HD bool  is_background(float depth, bool reversed) { return reversed ? depth < 1e-6f : depth >= (1.0f - 1e-6f); } // SSAO_Common.fxh:16-23, SSR_Common.fxh:48-55
MIFX_HD float luminance601(v3 c) { return dot(c, v3{0.299f, 0.587f, 0.114f}); } // PostFX_Common.fxh:40
MIFX_HD float spatial_weight(float d, float sigma) { return m_exp(fdiv(-d, 2.0f * sigma * sigma)); } // PostFX_Common.fxh:134
// the same weight for compile-time constant arguments (Poisson-disk radii in unrolled loops): plain expf / division so that the compiler folds it
MIFX_HD float spatial_weight_const(float d, float sigma) { return expf(-d / (2.0f * sigma * sigma)); }
// PostFX_Common.fxh:57-65
MIFX_HD float bayer4x4(uint32_t px, uint32_t py, uint32_t frame)
{
    uint32_t wx = px & 3u, wy = py & 3u;
    uint32_t A = 2068378560u * (1u - (wx >> 1u)) + 1500172770u * (wx >> 1u);
    uint32_t B = (wy + ((wx & 1u) << 2u)) << 2u;
    uint32_t bayer = ((A >> B) + frame) & 0xFu;
    return float(bayer) / 16.0f;
}
MIFX_HD v2 rotate_vector(v4 r, v2 v) { return v2{v.x * r.x + v.y * r.y, v.x * r.z + v.y * r.w}; } // RotateVector: Vec.x*Rotator.xz + Vec.y*Rotator.yw

// ------------------------------------------------------------------------------------------------ pitched image views
struct Img
{
    unsigned char* p;
    int w, h, pitch;
    int y0, yn; // row window [y0, y0 + yn) that a launch writing this image covers (yn == 0: all rows) -- row-band sharding, DESIGN.md section 6
};
MIFX_HD int row_end(const Img& o) { return o.yn ? o.y0 + o.yn : o.h; }
// Planes always live in HBM: loads / stores name the global address space, so that a descriptor fetched from LDS (stage_pyramid) does not
// degrade them to flat accesses.
#define MIFX_GLOBAL __attribute__((address_space(1)))
typedef float mifx_f2 __attribute__((ext_vector_type(2)));
typedef float mifx_f4 __attribute__((ext_vector_type(4)));
template <class T> struct GlobalAccess;
template <> struct GlobalAccess<float>
{
    static MIFX_D float load(const unsigned char* p) { return *(const MIFX_GLOBAL float*)p; }
    static MIFX_D void  store(unsigned char* p, float v) { *(MIFX_GLOBAL float*)p = v; }
};
template <> struct GlobalAccess<v2>
{
    static MIFX_D v2   load(const unsigned char* p) { const mifx_f2 t = *(const MIFX_GLOBAL mifx_f2*)p; return v2{t.x, t.y}; }
    static MIFX_D void store(unsigned char* p, v2 v) { *(MIFX_GLOBAL mifx_f2*)p = mifx_f2{v.x, v.y}; }
};
// The 4-channel texel in memory.  fp32 build: float4 (16 bytes).  -DMIFX_STORAGE_H4 (libmifx_h4.so): RGBA16_FLOAT (8 bytes) -- a load widens, a store rounds
// to nearest-even binary16 (v_cvt_f16_f32, overflow -> infinity: the conversion of a D3D / Vulkan RGBA16_FLOAT render-target write); registers and LDS stay fp32.
#ifdef MIFX_STORAGE_H4
typedef _Float16 mifx_h4 __attribute__((ext_vector_type(4)));
constexpr unsigned kV4Bytes = 8;
MIFX_HD v4 quantize_v4(v4 v) { return v4{float(_Float16(v.x)), float(_Float16(v.y)), float(_Float16(v.z)), float(_Float16(v.w))}; } // what a store + load does to a value
template <> struct GlobalAccess<v4>
{
    static MIFX_D v4   load(const unsigned char* p) { const mifx_h4 t = *(const MIFX_GLOBAL mifx_h4*)p; return v4{float(t.x), float(t.y), float(t.z), float(t.w)}; }
    static MIFX_D void store(unsigned char* p, v4 v) { *(MIFX_GLOBAL mifx_h4*)p = mifx_h4{_Float16(v.x), _Float16(v.y), _Float16(v.z), _Float16(v.w)}; }
};
#else
constexpr unsigned kV4Bytes = 16;
MIFX_HD v4 quantize_v4(v4 v) { return v; }
template <> struct GlobalAccess<v4>
{
    static MIFX_D v4   load(const unsigned char* p) { const mifx_f4 t = *(const MIFX_GLOBAL mifx_f4*)p; return v4{t.x, t.y, t.z, t.w}; }
    static MIFX_D void store(unsigned char* p, v4 v) { *(MIFX_GLOBAL mifx_f4*)p = mifx_f4{v.x, v.y, v.z, v.w}; }
};
#endif
template <class T> struct TexelBytes { static constexpr unsigned value = sizeof(T); };
template <> struct TexelBytes<v4> { static constexpr unsigned value = kV4Bytes; };
template <class T> struct Stored { using value = T; }; // the value a load of a T-texel returns / a store takes (the narrow storage types below widen to float / v2 / v4)

#include "mifx_ufloat.h" // float_to_ufloat / ufloat_to_float / quantize_ufloat: the unsigned small floats of R11G11B10_FLOAT

// ------------------------------------------------------------------------------------------------ per-plane storage types
// The effects name the texel type of every intermediate plane that the reference keeps in a narrow format:
//     ao_t     ambient occlusion (occlusion, accumulated, convoluted, resampled, output, history)  R8_UNORM    ScreenSpaceAmbientOcclusion.hpp:255
//     hl_t     SSAO history length                                                               R16_FLOAT   ScreenSpaceAmbientOcclusion.hpp:256
//     rough_t  SSR roughness                                                                     R8_UNORM    ScreenSpaceReflection.cpp:155
//     mask_t   SSR reflection mask (0 / 1; a D16 depth-stencil target in the reference)           one byte    ScreenSpaceReflection.cpp:47-51
//     var_t    SSR variance (resolved, history) and resolved depth                                R16_FLOAT   ScreenSpaceReflection.cpp:236, 247, 275
//     cm_t     closest motion                                                                    RG16_FLOAT  PostFXContext.cpp:281
//     bloom_t  Bloom pyramid levels and output                                                   R11G11B10_FLOAT  Bloom.cpp:111, 125, 137
//     coc_t    depth of field: signed circle of confusion, its temporal history                   R16_FLOAT   DepthOfField.cpp:196-223
//     dil_t    depth of field: separated / dilated / blurred near-field circle of confusion       R16_UNORM   DepthOfField.cpp:227-253
// fp32 build: float / float2 / float4 like every other plane.  Native-storage build (-DMIFX_STORAGE_H4): the reference's formats -- a load widens, a store
// converts the way a render-target write of that format does (UNORM: clamp, scale, + 0.5, truncate; FLOAT: round to nearest even; R11G11B10: no sign, no alpha).
#ifdef MIFX_STORAGE_H4
struct st_unorm8 {};
struct st_half {};
struct st_half2 {};
struct st_r11g11b10 {};
struct st_unorm16 {};
template <> struct Stored<st_unorm16> { using value = float; };
template <> struct TexelBytes<st_unorm16> { static constexpr unsigned value = 2; };
template <> struct Stored<st_unorm8> { using value = float; };
template <> struct Stored<st_half> { using value = float; };
template <> struct Stored<st_half2> { using value = v2; };
template <> struct Stored<st_r11g11b10> { using value = v4; };
template <> struct TexelBytes<st_unorm8> { static constexpr unsigned value = 1; };
template <> struct TexelBytes<st_half> { static constexpr unsigned value = 2; };
template <> struct TexelBytes<st_half2> { static constexpr unsigned value = 4; };
template <> struct TexelBytes<st_r11g11b10> { static constexpr unsigned value = 4; };
template <> struct GlobalAccess<st_unorm8>
{
    // c / 255 correctly rounded without a division: q = c * fl(1 / 255) and one residual step (verified for all 256 codes)
    static MIFX_D float load(const unsigned char* p)
    {
        const float c = float(*(const MIFX_GLOBAL unsigned char*)p), r = 1.0f / 255.0f, q = c * r;
        return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, c), r, q);
    }
    static MIFX_D void store(unsigned char* p, float v)
    {
        v = v != v ? 0.0f : fminf(fmaxf(v, 0.0f), 1.0f);
        float s = v * 255.0f;
        asm volatile("" : "+v"(s)); // (the product is rounded before the addition in every translation unit: a source compiled with -ffp-contract=fast would fuse the two)
        *(MIFX_GLOBAL unsigned char*)p = (unsigned char)(unsigned(s + 0.5f));
    }
};
// R16_UNORM, as st_unorm8: c / 65535 correctly rounded as q = c * fl(1 / 65535) and one residual step (all 65536 codes: tests/test_formats.py)
MIFX_D float unorm16_value(float c)
{
    const float r = 1.0f / 65535.0f, q = c * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 65535.0f, c), r, q);
}
MIFX_D float unorm16_code(float v)
{
    v = v != v ? 0.0f : fminf(fmaxf(v, 0.0f), 1.0f);
    float s = v * 65535.0f;
    asm volatile("" : "+v"(s)); // (rounded before the addition in every translation unit, as for st_unorm8)
    return float(unsigned(s + 0.5f));
}
template <> struct GlobalAccess<st_unorm16>
{
    static MIFX_D float load(const unsigned char* p) { return unorm16_value(float(*(const MIFX_GLOBAL unsigned short*)p)); }
    static MIFX_D void  store(unsigned char* p, float v) { *(MIFX_GLOBAL unsigned short*)p = (unsigned short)(unsigned(unorm16_code(v))); }
};
template <> struct GlobalAccess<st_half>
{
    static MIFX_D float load(const unsigned char* p) { return float(*(const MIFX_GLOBAL _Float16*)p); }
    static MIFX_D void  store(unsigned char* p, float v) { *(MIFX_GLOBAL _Float16*)p = _Float16(v); }
};
typedef _Float16 mifx_h2 __attribute__((ext_vector_type(2)));
template <> struct GlobalAccess<st_half2>
{
    static MIFX_D v2   load(const unsigned char* p) { const mifx_h2 t = *(const MIFX_GLOBAL mifx_h2*)p; return v2{float(t.x), float(t.y)}; }
    static MIFX_D void store(unsigned char* p, v2 v) { *(MIFX_GLOBAL mifx_h2*)p = mifx_h2{_Float16(v.x), _Float16(v.y)}; }
};
MIFX_D unsigned pack_r11g11b10(v4 v) // == float_to_ufloat per channel (tools/check_ufloat.cpp), a third of the instructions
{
    return encode_quantized<6>(quantize_ufloat<6>(v.x)) | (encode_quantized<6>(quantize_ufloat<6>(v.y)) << 11) | (encode_quantized<5>(quantize_ufloat<5>(v.z)) << 22);
}
MIFX_D v4 unpack_r11g11b10(unsigned t) { return v4{ufloat_to_float<6>(t & 0x7ffu), ufloat_to_float<6>((t >> 11) & 0x7ffu), ufloat_to_float<5>(t >> 22), 1.0f}; } // no alpha channel: reads as 1
template <> struct GlobalAccess<st_r11g11b10>
{
    static MIFX_D v4   load(const unsigned char* p) { return unpack_r11g11b10(*(const MIFX_GLOBAL unsigned*)p); }
    static MIFX_D void store(unsigned char* p, v4 v) { *(MIFX_GLOBAL unsigned*)p = pack_r11g11b10(v); }
};
typedef st_unorm8 ao_t;
typedef st_unorm8 rough_t;
typedef st_unorm8 mask_t; // (0 and 1 are exact in R8_UNORM)
typedef st_half hl_t;
typedef st_half var_t;
typedef st_half2 cm_t;
typedef st_r11g11b10 bloom_t;
typedef st_half coc_t;
typedef st_unorm16 dil_t;
// what a store + load of a Bloom texel does to a value (quantize_ufloat == decode(encode()) for every float: tools/check_ufloat.cpp)
MIFX_D v4 quantize_bloom(v4 v) { return v4{quantize_ufloat<6>(v.x), quantize_ufloat<6>(v.y), quantize_ufloat<5>(v.z), 1.0f}; }
#else
typedef float ao_t;
typedef float rough_t;
typedef float mask_t;
typedef float hl_t;
typedef float var_t;
typedef v2 cm_t;
typedef v4 bloom_t;
typedef float coc_t;
typedef float dil_t;
MIFX_HD v4 quantize_bloom(v4 v) { return v; }
#endif
// what a store followed by a load of a T-texel does to a value (identity for the full-precision types)
template <class T> MIFX_D float quantize_as(float v) { return v; }
#ifdef MIFX_STORAGE_H4
template <> MIFX_D float quantize_as<st_unorm8>(float v)
{
    v = v != v ? 0.0f : fminf(fmaxf(v, 0.0f), 1.0f);
    float s = v * 255.0f;
    asm volatile("" : "+v"(s)); // (as GlobalAccess<st_unorm8>::store)
    const float c = float(unsigned(s + 0.5f)), r = 1.0f / 255.0f, q = c * r;
    return __builtin_fmaf(__builtin_fmaf(-q, 255.0f, c), r, q);
}
template <> MIFX_D float quantize_as<st_half>(float v) { return float(_Float16(v)); }
template <> MIFX_D float quantize_as<st_unorm16>(float v) { return unorm16_value(unorm16_code(v)); }
#endif
// FEATURE_FLAG_HALF_PRECISION_DEPTH (PostFXContext.cpp:259-270, ScreenSpaceAmbientOcclusion.cpp:95-97): the reference allocates the reprojected / previous depth and SSAO's two depth
// pyramids as R16_UNORM.  The flag is a run-time switch and the texel type of a plane a compile-time one, so the native-storage build gives those planes the VALUES an
// R16_UNORM target keeps, in 4-byte texels; the fp32 build stores full precision like everywhere else.
MIFX_D float depth16(float v, int on)
{
#ifdef MIFX_STORAGE_H4
    return on ? quantize_as<st_unorm16>(v) : v;
#else
    (void)on;
    return v;
#endif
}
template <class T> MIFX_D typename Stored<T>::value ld(const Img& im, int x, int y) { return GlobalAccess<T>::load(im.p + size_t(y) * im.pitch + size_t(x) * TexelBytes<T>::value); }
template <class T> MIFX_D void st(const Img& im, int x, int y, typename Stored<T>::value v) { GlobalAccess<T>::store(im.p + size_t(y) * im.pitch + size_t(x) * TexelBytes<T>::value, v); }
// A texel that is read ONCE in the whole launch (the thread's own texel of a plane nobody else touches): loaded with the non-temporal hint, so that the line is not kept in
// the L2 in front of lines that will be asked for again (-DMIFX_NT_LOADS=0: plain loads, for A/B builds).  Planes that neighbouring threads read as well -- filter taps,
// bilinear footprints, tiles -- keep plain loads: with the hint on the history taps of TAA and R6 (planes those passes are the last readers of) TAA takes 268.7 instead of 162.3 us
// and R6 116.8 instead of 98.8 -- the neighbours' re-reads of a line miss.
#ifndef MIFX_NT_LOADS
#define MIFX_NT_LOADS 1
#endif
template <class T> struct StreamAccess
{
    static MIFX_D typename Stored<T>::value load(const unsigned char* p) { return GlobalAccess<T>::load(p); }
};
#if defined(__HIP_DEVICE_COMPILE__) && MIFX_NT_LOADS
template <> struct StreamAccess<float>
{
    static MIFX_D float load(const unsigned char* p) { return __builtin_nontemporal_load((const MIFX_GLOBAL float*)p); }
};
template <> struct StreamAccess<v2>
{
    static MIFX_D v2 load(const unsigned char* p) { const mifx_f2 t = __builtin_nontemporal_load((const MIFX_GLOBAL mifx_f2*)p); return v2{t.x, t.y}; }
};
template <> struct StreamAccess<v4>
{
#ifdef MIFX_STORAGE_H4
    static MIFX_D v4 load(const unsigned char* p) { const mifx_h4 t = __builtin_nontemporal_load((const MIFX_GLOBAL mifx_h4*)p); return v4{float(t.x), float(t.y), float(t.z), float(t.w)}; }
#else
    static MIFX_D v4 load(const unsigned char* p) { const mifx_f4 t = __builtin_nontemporal_load((const MIFX_GLOBAL mifx_f4*)p); return v4{t.x, t.y, t.z, t.w}; }
#endif
};
#endif
// ... and non-temporal STORES for the two 4-channel targets that nothing reads before the cache has turned over several times (MIFX_NT_STORES, one bit each: 1 = the LDR frame
// of Bloom's final pass, which the chain never reads; 2 = the shade's radiance and specular-IBL targets, whose next full reader, the composite, runs a millisecond and ~5 GB of
// traffic later -- the march's loads of single hit texels do not live on cached lines either way).  Their 400 MB per frame no longer push out lines that will be asked for:
// one stream Bloom's final pass 80.6 -> 76.8 us, the PostFX prep behind the shade 50.5 -> 47.5, R4 305.5 -> 299.2; the three-lane frame 1.6196 -> 1.6018 ms (-1.1 %, three
// repetitions each, same box; the LDR frame alone -0.4 %) -- profiles/r06_ab_rows_up.txt.  Every other store of the frame feeds the next pass and stays plain: a non-temporal
// store costs the consumer its cache hits (tools/microbench/mall_reuse.hip: 30.6 -> 39.6 us).  (Measured and not taken: the pixel's own NORMAL texel in R4 / R5 / A3 loaded
// non-temporally -- three VALU-bound readers of one plane at three times of the frame: 1.6002 against 1.6018 ms, nothing.)
#ifndef MIFX_NT_STORES
#define MIFX_NT_STORES 3
#endif
template <int BIT> MIFX_D void st_v4_late(const Img& im, int x, int y, v4 v)
{
#if defined(__HIP_DEVICE_COMPILE__)
    if (MIFX_NT_STORES & BIT)
    {
        unsigned char* p = im.p + size_t(y) * im.pitch + size_t(x) * kV4Bytes;
#ifdef MIFX_STORAGE_H4
        __builtin_nontemporal_store(mifx_h4{_Float16(v.x), _Float16(v.y), _Float16(v.z), _Float16(v.w)}, (MIFX_GLOBAL mifx_h4*)p);
#else
        __builtin_nontemporal_store(mifx_f4{v.x, v.y, v.z, v.w}, (MIFX_GLOBAL mifx_f4*)p);
#endif
        return;
    }
#endif
    st<v4>(im, x, y, v);
}
template <class T> MIFX_D typename Stored<T>::value ld_once(const Img& im, int x, int y) { return StreamAccess<T>::load(im.p + size_t(y) * im.pitch + size_t(x) * TexelBytes<T>::value); }
template <class T> MIFX_D typename Stored<T>::value ld_clamp(const Img& im, int x, int y) { return ld<T>(im, clampi(x, 0, im.w - 1), clampi(y, 0, im.h - 1)); }
// D3D Load semantics: out-of-bounds returns 0
template <class T = float> MIFX_D float ld_zero_f(const Img& im, int x, int y) { return (x < 0 || y < 0 || x >= im.w || y >= im.h) ? 0.0f : ld<T>(im, x, y); }
template <class T = v2> MIFX_D v2 ld_zero_v2(const Img& im, int x, int y) { return (x < 0 || y < 0 || x >= im.w || y >= im.h) ? v2{0.f, 0.f} : ld<T>(im, x, y); }

// Thread -> pixel mapping for divergent / gather-heavy kernels: one wave covers an 8x8 pixel tile instead of a 64x1 strip (coherent rays and
// scattered taps, better L1 locality, whole tiles of masked-out pixels retire at once); a 256-thread block covers 32x8 pixels.
// Launch with block (256,1,1) and grid ((w+31)/32, (h+7)/8).
// (An XCD-contiguous remap of the workgroup index -- XCD k <- the k-th eighth of the image -- was measured and rejected: +35 % on R4 and
//  +20 % on A3, because the work per block is very uneven (sky / masked-out regions) and the round-robin placement balances it.  Round 3: the balanced
//  form of the same idea -- super-blocks of 16 x 8 workgroups, every XCD walks its own 4 x 4 chunk (128 x 32 pixels) of each -- is +3.5 % on R4 and +9 % on
//  A3 (profiles/r03_ab_xcd_chunks.txt): the L2s are not what these kernels wait for.)
MIFX_D bool tiled_xy(const Img& out, int& x, int& y) // false: outside the image / the row window of `out`
{
    const int t = threadIdx.x, lane = t & 63;
    x = int(blockIdx.x) * int(blockDim.x >> 3) + (t >> 6) * 8 + (lane & 7); // (256 threads: 32 pixels per block row; 64 threads: one 8x8 tile per block)
    y = int(blockIdx.y) * 8 + (lane >> 3) + out.y0;
    return x < out.w && y < row_end(out);
}
// EXPERIMENT (round 6, -DMIFX_XCD_COLUMNS=1; R4 and A3 only): the workgroups of a grid row go to the eight XCDs round-robin (workgroup i of a row whose length is a multiple
// of eight runs on XCD i % 8), so with the plain mapping every XCD -- every L2 -- serves 32-pixel columns spread over the whole width.  Here XCD k gets the k-th eighth of every
// tile row instead: a contiguous column block per L2, all XCDs still walking down the image together (the vertical work profile -- sky above, reflective ground below -- is what
// unbalanced the "k-th eighth of the image" of round 1, and the 128 x 32 chunks of round 3 kept an L2's tiles apart in both directions).  The launcher rounds the grid's x up to a
// multiple of eight (xcd_grid_x); the tiles beyond the image fall out at the bounds test.
// MEASURED AND NOT TAKEN (profiles/r06_ab_xcd_columns.txt): R4 309.1 -> 343.5 us, A3 209.8 -> 209.9 us, same box, bit-identical output.  A column block per XCD unbalances
// the march as well (an eighth of the width is one or two spheres wide), and A3 does not notice where its L2 lines come from: the third form of this idea, the third loss.
// (Also measured: the tile rows dispatched from the last to the first -- the reflective ground before the sky -- R4 310.1 -> 316.7 us, A3 209.1 -> 205.9 us: nothing.)
#ifndef MIFX_XCD_COLUMNS
#define MIFX_XCD_COLUMNS 0
#endif
MIFX_D bool tiled_xy_xcd(const Img& out, int& x, int& y)
{
#if MIFX_XCD_COLUMNS
    const int t = threadIdx.x, lane = t & 63;
    const int perXcd = int(gridDim.x) >> 3, b = int(blockIdx.x), bx = (b & 7) * perXcd + (b >> 3);
    x = bx * int(blockDim.x >> 3) + (t >> 6) * 8 + (lane & 7);
    y = int(blockIdx.y) * 8 + (lane >> 3) + out.y0;
    return x < out.w && y < row_end(out);
#else
    return tiled_xy(out, x, y);
#endif
}
MIFX_HD unsigned xcd_grid_x(unsigned gx) { return MIFX_XCD_COLUMNS ? (gx + 7u) / 8u * 8u : gx; }
MIFX_D bool tiled_xy_at(const Img& out, int bx, int& x, int& y) // the same with the workgroup's column given (a grid whose workgroups do not all compute pixels)
{
    const int t = threadIdx.x, lane = t & 63;
    x = bx * int(blockDim.x >> 3) + (t >> 6) * 8 + (lane & 7);
    y = int(blockIdx.y) * 8 + (lane >> 3) + out.y0;
    return x < out.w && y < row_end(out);
}
// linear mapping: block (bx, by) covers bx x by pixels of the row window of `out`
MIFX_D bool pixel_xy(const Img& out, int& x, int& y)
{
    x = int(blockIdx.x * blockDim.x + threadIdx.x);
    y = int(blockIdx.y * blockDim.y + threadIdx.y) + out.y0;
    return x < out.w && y < row_end(out);
}

// Round 6: WHICH WAY A PASS WALKS ITS ROWS (MIFX_ROWS_UP, one bit per pass; 0 = every pass from the first row to the last, as until now).  The dispatcher hands out workgroups in
// grid order, so a pass reads and writes its planes from the top of the image to the bottom -- and every large pass of the frame moves 0.5 - 0.85 GB through a 256 MB Infinity
// Cache: when the next pass starts at the top, the rows it asks for first were evicted long ago, and the rows the cache still holds -- the LAST ones written -- are the ones
// it reaches last, when they are gone too.  A pass that walks the other way starts where its producer stopped.  So the passes of a lane alternate wherever one consumes a
// large plane of the one in front of it:  R4 down, R5 UP (R4's two ray planes), R6 down (R5's three), the composite UP (R6's history), TAA down (the composite), Bloom's prefilter
// UP (TAA's output), the first down-sampling down; and on the SSAO lane A3 UP (the pyramid A2 has just written), A5 UP (its work lists then hand A8 the rows A5 wrote last).
// Same texels, same arithmetic: the mapping of workgroups to rows is all that changes (bit-identical; the GPU suite runs on this build).
// Measured (profiles/r06_ab_rows_up.txt): one stream, per launch -- composite 168.3 -> 159.4 us, R5 131.5 -> 126.9, R6 106.8 -> 103.1, A8 (list) 55.6 -> 51.3, A3 210.8 -> 208.6,
// A7 (list) 44.1 -> 45.3: the sum of the frame's kernels -1.3 %; the three-lane frame -0.8 .. -1.1 % on three boxes (1.6554 -> 1.6386 ms on the slowest).  The control with
// R6 and TAA turned as well (neighbours walking the same way again) gives it all back.  Not taken: the shade up (-2 us, the march beside it +2), the PostFX prep up (nothing), Bloom's
// final pass up (nothing: TAA's whole output is still in the cache when it runs).
//   bits: 1 = R5, 2 = the composite, 4 = Bloom's prefilter, 32 = A5, 1024 = A3
#ifndef MIFX_ROWS_UP
#define MIFX_ROWS_UP 1063
#endif
template <int BIT> MIFX_D unsigned block_row() { return (MIFX_ROWS_UP & BIT) ? gridDim.y - 1u - blockIdx.y : blockIdx.y; }
template <int BIT> MIFX_D bool pixel_xy_dir(const Img& out, int& x, int& y)
{
    x = int(blockIdx.x * blockDim.x + threadIdx.x);
    y = int(block_row<BIT>() * blockDim.y + threadIdx.y) + out.y0;
    return x < out.w && y < row_end(out);
}

// "These fetched values are needed HERE": an empty asm that takes the registers as in/out operands.  A group of independent loads written back to back and followed
// by one keep_here() per value is issued together and waited for once; without it the compiler sinks a load whose value is only used behind a later branch into that
// branch (each then costs its own round trip: measured on the latency-bound passes in round 3).
MIFX_D void keep_here(float& a) { asm volatile("" : "+v"(a)); }
MIFX_D void keep_here(v2& a) { asm volatile("" : "+v"(a.x), "+v"(a.y)); }
MIFX_D void keep_here(v4& a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }

// mip chain of a single-channel or float4 pyramid (tightly described by per-level views)
struct Pyr
{
    Img l[8];
    int levels;
};

// Mip selection by a per-lane level would index the kernel-argument copy of a Pyr with a VGPR: the compiler turns that into dependent
// vector loads from the kernarg segment (size, then pointer / pitch, then the texel: three memory round trips per sample).  Kernels that
// sample a pyramid at divergent levels copy the level descriptors to LDS once per block instead; the lookup is then a ds_read.
// Call from every thread of the block before any early return.
MIFX_D void stage_pyramid(Img* lds, const Pyr& p)
{
    const unsigned t = threadIdx.y * blockDim.x + threadIdx.x;
    if (t < 8u) lds[t] = p.l[t];
    __syncthreads();
}
// D3D Load semantics without a branch around the load: out-of-bounds lanes fetch texel (0, 0) and select 0
MIFX_D float ld_zero_f_nb(const Img& im, int x, int y)
{
    const bool  in = unsigned(x) < unsigned(im.w) && unsigned(y) < unsigned(im.h);
    const float v  = ld<float>(im, in ? x : 0, in ? y : 0);
    return in ? v : 0.0f;
}

// bilinear sampling info, unnormalised coords -- ShaderUtilities.fxh:126-142
struct Bilinear
{
    int   x0, y0, x1, y1;
    float w00, w10, w01, w11;
};
MIFX_HD Bilinear bilinear_uc(float lx, float ly, int w, int h)
{
    lx -= 0.5f; ly -= 0.5f;
    float fx = floorf(lx), fy = floorf(ly);
    Bilinear b;
    b.x0 = clampi(int(fx), 0, w - 1);
    b.y0 = clampi(int(fy), 0, h - 1);
    b.x1 = clampi(int(fx) + 1, 0, w - 1);
    b.y1 = clampi(int(fy) + 1, 0, h - 1);
    float x = lx - fx, y = ly - fy;
    b.w00 = (1.0f - x) * (1.0f - y);
    b.w10 = x * (1.0f - y);
    b.w01 = (1.0f - x) * y;
    b.w11 = x * y;
    return b;
}
// int(floorf(x)) in one instruction (v_cvt_flr_i32_f32; the compiler only selects it under unsafe-fp-math)
MIFX_D int floor_to_int(float x)
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    __asm__("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
#else // (a host compilation of the kernel headers: tests/host_kernels)
    return int(floorf(x));
#endif
}
// ---- bilinear taps with 32-bit texel offsets.  int(floor(x)) and x - floor(x) are one instruction each (v_cvt_flr_i32_f32, v_fract_f32: the
// same value as the subtraction except that a result that would round up to 1.0 stays just below it), the clamp of a texel index is one
// v_med3_i32, and the byte offset y * pitch + x * texel is a 24-bit multiply-add; added to the (uniform) plane pointer it selects the
// scalar-base form of the global load: a tap costs 62 vector instructions instead of 85 in the bokeh gather, TAA's five history taps got 4 %
// faster.  (The same 32-bit offsets in the plain ld / st helpers were measured twice and lose: +1 % over the chain, R5 / R6 / R7 +4-8 %.)  Planes are < 4 GiB and
// pitches / heights < 2^24 by construction (mifx_image2d: uint32 pitch, frames up to 16k x 16k float4).
MIFX_D int med3i(int x, int lo, int hi) // min(max(x, lo), hi) for lo <= hi
{
#if defined(__HIP_DEVICE_COMPILE__)
    int r;
    __asm__("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(lo), "v"(hi));
    return r;
#else
    return x < lo ? lo : x > hi ? hi : x;
#endif
}
// (the two device intrinsics of the addressing helpers, with the plain expressions for a host compilation of the kernel headers: tests/host_kernels)
#if defined(__HIP_DEVICE_COMPILE__)
MIFX_HD float    m_fract(float x) { return __builtin_amdgcn_fractf(x); }
MIFX_HD unsigned m_umul24(unsigned a, unsigned b) { return __umul24(a, b); }
#else
MIFX_HD float    m_fract(float x) { return fminf(x - floorf(x), 0x1.fffffep-1f); }
MIFX_HD unsigned m_umul24(unsigned a, unsigned b) { return a * b; }
#endif
MIFX_D unsigned texel_offset(const Img& im, int x, int y, unsigned texelBytes) { return m_umul24(unsigned(y), unsigned(im.pitch)) + unsigned(x) * texelBytes; }
template <class T> MIFX_D typename Stored<T>::value ld_at(const Img& im, unsigned byteOffset) { return GlobalAccess<T>::load(im.p + byteOffset); }
struct BilinearTaps
{
    unsigned o00, o10, o01, o11; // byte offsets of the four texels (clamp addressing)
    float    w00, w10, w01, w11; // GetBilinearSamplingInfoUC weights (ShaderUtilities.fxh:126-142)
};
template <unsigned TEXEL_BYTES> MIFX_D BilinearTaps bilinear_taps(const Img& im, float u, float v)
{
    const float lx = u * float(im.w) - 0.5f, ly = v * float(im.h) - 0.5f;
    const int   ix = floor_to_int(lx), iy = floor_to_int(ly);
    const float x = m_fract(lx), y = m_fract(ly);
    const int   x0 = med3i(ix, 0, im.w - 1), x1 = med3i(ix + 1, 0, im.w - 1), y0 = med3i(iy, 0, im.h - 1), y1 = med3i(iy + 1, 0, im.h - 1);
    const unsigned r0 = m_umul24(unsigned(y0), unsigned(im.pitch)), r1 = m_umul24(unsigned(y1), unsigned(im.pitch));
    const unsigned c0 = unsigned(x0) * TEXEL_BYTES, c1 = unsigned(x1) * TEXEL_BYTES;
    BilinearTaps b;
    b.o00 = r0 + c0; b.o10 = r0 + c1; b.o01 = r1 + c0; b.o11 = r1 + c1;
    b.w00 = (1.0f - x) * (1.0f - y); b.w10 = x * (1.0f - y); b.w01 = (1.0f - x) * y; b.w11 = x * y;
    return b;
}
MIFX_D v4 sample_linear_clamp_v4_taps(const Img& im, float u, float v)
{
    const BilinearTaps b = bilinear_taps<kV4Bytes>(im, u, v);
    const v4 t00 = ld_at<v4>(im, b.o00), t10 = ld_at<v4>(im, b.o10), t01 = ld_at<v4>(im, b.o01), t11 = ld_at<v4>(im, b.o11);
    {
        MIFX_FMA_BLOCK
        return v4{t00.x * b.w00 + t10.x * b.w10 + t01.x * b.w01 + t11.x * b.w11, t00.y * b.w00 + t10.y * b.w10 + t01.y * b.w01 + t11.y * b.w11,
                  t00.z * b.w00 + t10.z * b.w10 + t01.z * b.w01 + t11.z * b.w11, t00.w * b.w00 + t10.w * b.w10 + t01.w * b.w01 + t11.w * b.w11};
    }
}
// An HDR frame as the tone map and the auto exposure take it: the 4-channel colour texel, or -- native-storage build, `packed` -- Bloom's own R11G11B10_FLOAT output plane
// (Bloom.cpp:137; no alpha channel: reads as 1).
MIFX_D v4 ld_hdr(const Img& im, int x, int y, int packed)
{
#ifdef MIFX_STORAGE_H4
    if (packed) return ld<st_r11g11b10>(im, x, y);
#else
    (void)packed;
#endif
    return ld<v4>(im, x, y);
}
MIFX_D v4 ld_hdr_once(const Img& im, int x, int y, int packed) // (the thread's own texel of a frame nothing else of the launch reads: ld_once)
{
#ifdef MIFX_STORAGE_H4
    if (packed) return ld<st_r11g11b10>(im, x, y);
#else
    (void)packed;
#endif
    return ld_once<v4>(im, x, y);
}
MIFX_D v4 sample_linear_clamp_hdr(const Img& im, float u, float v, int packed)
{
#ifdef MIFX_STORAGE_H4
    if (packed)
    {
        const BilinearTaps b = bilinear_taps<4u>(im, u, v);
        const v4 t00 = ld_at<st_r11g11b10>(im, b.o00), t10 = ld_at<st_r11g11b10>(im, b.o10), t01 = ld_at<st_r11g11b10>(im, b.o01), t11 = ld_at<st_r11g11b10>(im, b.o11);
        {
            MIFX_FMA_BLOCK
            return v4{t00.x * b.w00 + t10.x * b.w10 + t01.x * b.w01 + t11.x * b.w11, t00.y * b.w00 + t10.y * b.w10 + t01.y * b.w01 + t11.y * b.w11,
                      t00.z * b.w00 + t10.z * b.w10 + t01.z * b.w01 + t11.z * b.w11, 1.0f};
        }
    }
#else
    (void)packed;
#endif
    return sample_linear_clamp_v4_taps(im, u, v);
}
template <class T = float> MIFX_D float sample_linear_clamp_f_taps(const Img& im, float u, float v)
{
    const BilinearTaps b = bilinear_taps<TexelBytes<T>::value>(im, u, v);
    return ld_at<T>(im, b.o00) * b.w00 + ld_at<T>(im, b.o10) * b.w10 + ld_at<T>(im, b.o01) * b.w01 + ld_at<T>(im, b.o11) * b.w11;
}
// SampleLevel with a linear-clamp sampler at normalised uv (single-channel plane of storage type T)
template <class T = float> MIFX_D float sample_linear_clamp_f(const Img& im, float u, float v) { return sample_linear_clamp_f_taps<T>(im, u, v); }
MIFX_D v4    sample_linear_clamp_v4(const Img& im, float u, float v) { return sample_linear_clamp_v4_taps(im, u, v); }
// SampleLevel with a point-clamp sampler
template <class T = float> MIFX_D float sample_point_clamp_f(const Img& im, float u, float v)
{
    int x = clampi(floor_to_int(u * float(im.w)), 0, im.w - 1);
    int y = clampi(floor_to_int(v * float(im.h)), 0, im.h - 1);
    return ld<T>(im, x, y);
}

} // namespace mifx
