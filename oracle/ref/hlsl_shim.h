// hlsl_shim.h -- TEST INFRASTRUCTURE ONLY (never linked into the product library).
//
// A C++20 emulation of the subset of HLSL that the DiligentFX hot-path shaders use, so that the
// reference's own .fx/.fxh source (read from /root/reference at build time, never copied into this
// repository) compiles with g++ and can be executed pixel-by-pixel on the host.  The result,
// oracle/_ref/libmifx_ref.so, is the *reference itself* running on the CPU and is what pins the
// hand-written oracle (oracle/mifx_oracle.cpp) and, through it, the HIP kernels.
//
// What is emulated (see SURVEY.md Appendix A for the DiligentCore conventions fixed here):
//   * floatN/intN/uintN with full xyzw/rgba swizzles (read + write), float3x3/float4x4, mul()
//   * the intrinsic set used by the shaders (componentwise math, dot/cross/normalize/reflect, asuint, ...)
//   * Texture2D<T>/TextureCube with Load / Sample / SampleLevel / GetDimensions and SamplerState
//     (point|linear min/mag, point|linear mip, clamp|border|wrap), exact fp32 filter weights as
//     in GetBilinearSamplingInfoUC (Shaders/Common/public/ShaderUtilities.fxh:126-142)
//   * cbuffer, discard, ddx/ddy (2x2 quad two-phase emulation), SV_* semantics (stripped by ref_prep.py)
//   * DiligentCore platform macros for the D3D/Vulkan path: NDC_MIN_Z=0, UV y-down, row-vector matrices.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <cfloat>
#include <algorithm>
#include <type_traits>
#include <vector>

#undef M_PI
#undef FLT_MAX
#undef FLT_MIN
#undef FLT_EPS

namespace hlsl
{
typedef unsigned int uint;

template <class T, int N> struct vec;
template <class V, class T, int N, int... I> struct swz;

// ----------------------------------------------------------------------------------------------
// component-count trait + flatten helper for the generic constructor
template <class A, class = void> struct ncomp { static constexpr int value = -100000; };
template <class A> struct ncomp<A, std::enable_if_t<std::is_arithmetic_v<A>>> { static constexpr int value = 1; };
template <class T, int N> struct ncomp<vec<T, N>> { static constexpr int value = N; };
template <class V, class T, int N, int... I> struct ncomp<swz<V, T, N, I...>> { static constexpr int value = int(sizeof...(I)); };

template <class T, class A> inline std::enable_if_t<std::is_arithmetic_v<A>> hl_put(T*& o, const A& a) { *o++ = T(a); }
template <class T, class U, int M> inline void hl_put(T*& o, const vec<U, M>& a);
template <class T, class V, class U, int M, int... I> inline void hl_put(T*& o, const swz<V, U, M, I...>& a);

// ----------------------------------------------------------------------------------------------
// swizzle proxy: lives in a union with the vector storage
template <class V, class T, int N, int... I> struct swz
{
    T d[N];
    static constexpr int K = int(sizeof...(I));
    operator V() const { return V(d[I]...); }
    swz& operator=(const V& v)
    {
        constexpr int idx[] = {I...};
        for (int k = 0; k < K; ++k) d[idx[k]] = v.d[k];
        return *this;
    }
    swz& operator=(const swz& o)
    {
        V v = o;
        return *this = v;
    }
    template <class V2, int N2, int... J> swz& operator=(const swz<V2, T, N2, J...>& o)
    {
        V v = o;
        return *this = v;
    }
#define HL_SWZ_OP(op)                                       \
    swz& operator op##=(const V& v)                         \
    {                                                       \
        constexpr int idx[] = {I...};                       \
        for (int k = 0; k < K; ++k) d[idx[k]] op## = v.d[k]; \
        return *this;                                       \
    }                                                       \
    swz& operator op##=(T s)                                \
    {                                                       \
        constexpr int idx[] = {I...};                       \
        for (int k = 0; k < K; ++k) d[idx[k]] op## = s;      \
        return *this;                                       \
    }
    HL_SWZ_OP(+)
    HL_SWZ_OP(-)
    HL_SWZ_OP(*)
    HL_SWZ_OP(/)
#undef HL_SWZ_OP
};

// ----------------------------------------------------------------------------------------------
#define HL_VEC_COMMON(N)                                                                                  \
    vec() { for (int i = 0; i < N; ++i) d[i] = T(0); }                                                    \
    vec(const vec& o) { for (int i = 0; i < N; ++i) d[i] = o.d[i]; }                                      \
    vec& operator=(const vec& o) { for (int i = 0; i < N; ++i) d[i] = o.d[i]; return *this; }             \
    template <class... A, class = std::enable_if_t<(sizeof...(A) >= 1) && ((ncomp<A>::value + ...) == N)>> \
    explicit(sizeof...(A) == 1) vec(const A&... a) { T* o = d; (hl_put(o, a), ...); }                      \
    template <class A, class = std::enable_if_t<std::is_arithmetic_v<A> && (N > 1)>, class = void>        \
    explicit vec(A s) { for (int i = 0; i < N; ++i) d[i] = T(s); }                                        \
    T& operator[](int i) { return d[i]; }                                                                 \
    const T& operator[](int i) const { return d[i]; }                                                     \
    vec& operator+=(const vec& o) { for (int i = 0; i < N; ++i) d[i] += o.d[i]; return *this; }           \
    vec& operator-=(const vec& o) { for (int i = 0; i < N; ++i) d[i] -= o.d[i]; return *this; }           \
    vec& operator*=(const vec& o) { for (int i = 0; i < N; ++i) d[i] *= o.d[i]; return *this; }           \
    vec& operator/=(const vec& o) { for (int i = 0; i < N; ++i) d[i] /= o.d[i]; return *this; }           \
    vec& operator+=(T s) { for (int i = 0; i < N; ++i) d[i] += s; return *this; }                         \
    vec& operator-=(T s) { for (int i = 0; i < N; ++i) d[i] -= s; return *this; }                         \
    vec& operator*=(T s) { for (int i = 0; i < N; ++i) d[i] *= s; return *this; }                         \
    vec& operator/=(T s) { for (int i = 0; i < N; ++i) d[i] /= s; return *this; }

template <class T> struct vec<T, 2>
{
    union
    {
        T d[2];
        struct { T x, y; };
        struct { T r, g; };
#define HL_N 2
#include "hlsl_swizzles_2.inc"
#undef HL_N
    };
    HL_VEC_COMMON(2)
};
template <class T> struct vec<T, 3>
{
    union
    {
        T d[3];
        struct { T x, y, z; };
        struct { T r, g, b; };
#define HL_N 3
#include "hlsl_swizzles_3.inc"
#undef HL_N
    };
    HL_VEC_COMMON(3)
};
template <class T> struct vec<T, 4>
{
    union
    {
        T d[4];
        struct { T x, y, z, w; };
        struct { T r, g, b, a; };
#define HL_N 4
#include "hlsl_swizzles_4.inc"
#undef HL_N
    };
    HL_VEC_COMMON(4)
};

template <class T, class U, int M> inline void hl_put(T*& o, const vec<U, M>& a)
{
    for (int i = 0; i < M; ++i) *o++ = T(a.d[i]);
}
template <class T, class V, class U, int M, int... I> inline void hl_put(T*& o, const swz<V, U, M, I...>& a)
{
    constexpr int idx[] = {I...};
    for (int k = 0; k < int(sizeof...(I)); ++k) *o++ = T(a.d[idx[k]]);
}

typedef vec<float, 2> float2;
typedef vec<float, 3> float3;
typedef vec<float, 4> float4;
typedef vec<int, 2>   int2;
typedef vec<int, 3>   int3;
typedef vec<int, 4>   int4;
typedef vec<uint, 2>  uint2;
typedef vec<uint, 3>  uint3;
typedef vec<uint, 4>  uint4;

static_assert(sizeof(float2) == 8 && sizeof(float3) == 12 && sizeof(float4) == 16, "vector layout");

// ----------------------------------------------------------------------------------------------
// operators: NON-template overloads so that swizzle proxies convert implicitly
#define HL_BINOP(T, N, op)                                                                                                                         \
    inline vec<T, N> operator op(const vec<T, N>& a, const vec<T, N>& b) { vec<T, N> r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] op b.d[i]; return r; } \
    inline vec<T, N> operator op(const vec<T, N>& a, T b) { vec<T, N> r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] op b; return r; }                   \
    inline vec<T, N> operator op(T a, const vec<T, N>& b) { vec<T, N> r; for (int i = 0; i < N; ++i) r.d[i] = a op b.d[i]; return r; }
#define HL_ARITH(T, N) HL_BINOP(T, N, +) HL_BINOP(T, N, -) HL_BINOP(T, N, *) HL_BINOP(T, N, /) \
    inline vec<T, N> operator-(const vec<T, N>& a) { vec<T, N> r; for (int i = 0; i < N; ++i) r.d[i] = -a.d[i]; return r; } \
    inline vec<T, N> operator+(const vec<T, N>& a) { return a; }
#define HL_BITS(T, N) HL_BINOP(T, N, &) HL_BINOP(T, N, |) HL_BINOP(T, N, ^) HL_BINOP(T, N, <<) HL_BINOP(T, N, >>) HL_BINOP(T, N, %)
#define HL_FORALL_N(M, T) M(T, 2) M(T, 3) M(T, 4)
HL_FORALL_N(HL_ARITH, float)
HL_FORALL_N(HL_ARITH, int)
HL_FORALL_N(HL_ARITH, uint)
HL_FORALL_N(HL_BITS, int)
HL_FORALL_N(HL_BITS, uint)

// ----------------------------------------------------------------------------------------------
// scalar intrinsics
inline float abs(float x) { return std::fabs(x); }
inline int   abs(int x) { return x < 0 ? -x : x; }
inline float floor(float x) { return std::floor(x); }
inline float ceil(float x) { return std::ceil(x); }
inline float round(float x) { return std::nearbyint(x); }
inline float trunc(float x) { return std::trunc(x); }
inline float frac(float x) { return x - std::floor(x); }
inline float saturate(float x) { return std::fmin(std::fmax(x, 0.0f), 1.0f); } // D3D: saturate(NaN) = 0
inline float sqrt(float x) { return std::sqrt(x); }
inline float rsqrt(float x) { return 1.0f / std::sqrt(x); }
inline float exp(float x) { return std::exp(x); }
inline float exp2(float x) { return std::exp2(x); }
inline float log(float x) { return std::log(x); }
inline float log2(float x) { return std::log2(x); }
inline float log10(float x) { return std::log10(x); }
inline float sin(float x) { return std::sin(x); }
inline float cos(float x) { return std::cos(x); }
inline float tan(float x) { return std::tan(x); }
inline float asin(float x) { return std::asin(x); }
inline float acos(float x) { return std::acos(x); }
inline float atan(float x) { return std::atan(x); }
inline float rcp(float x) { return 1.0f / x; }
inline float sign(float x) { return x > 0.f ? 1.f : (x < 0.f ? -1.f : 0.f); }
// D3D / IEEE-754-2008 minNum/maxNum: if one operand is NaN the other is returned (the TAA AABB clip relies on it,
// TAA_ComputeTemporalAccumulation.fx:98-106 divides by a zero colour delta on static pixels)
inline float min(float a, float b) { return std::fmin(a, b); }
inline float max(float a, float b) { return std::fmax(a, b); }
inline int   min(int a, int b) { return a < b ? a : b; }
inline int   max(int a, int b) { return a > b ? a : b; }
inline uint  min(uint a, uint b) { return a < b ? a : b; }
inline uint  max(uint a, uint b) { return a > b ? a : b; }
inline float pow(float a, float b) { return std::pow(a, b); }
inline float step(float e, float x) { return x >= e ? 1.f : 0.f; }
inline float atan2(float y, float x) { return std::atan2(y, x); }
inline float fmod(float a, float b) { return std::fmod(a, b); }
inline float clamp(float x, float a, float b) { return min(max(x, a), b); }
inline int   clamp(int x, int a, int b) { return min(max(x, a), b); }
inline uint  clamp(uint x, uint a, uint b) { return min(max(x, a), b); }
inline float lerp(float a, float b, float t) { return a + t * (b - a); }
inline float smoothstep(float a, float b, float x)
{
    float t = saturate((x - a) / (b - a));
    return t * t * (3.0f - 2.0f * t);
}
inline void sincos(float a, float& s, float& c) { s = std::sin(a); c = std::cos(a); }
inline uint  asuint(float f) { uint u; std::memcpy(&u, &f, 4); return u; }
inline uint  asuint(uint u) { return u; }
inline int   asint(float f) { int u; std::memcpy(&u, &f, 4); return u; }
inline float asfloat(uint u) { float f; std::memcpy(&f, &u, 4); return f; }
inline float asfloat(int u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint  reversebits(uint v)
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0F0F0F0Fu) | ((v & 0x0F0F0F0Fu) << 4);
    v = ((v >> 8) & 0x00FF00FFu) | ((v & 0x00FF00FFu) << 8);
    return (v >> 16) | (v << 16);
}
inline uint countbits(uint v) { return uint(__builtin_popcount(v)); }
inline bool isnan(float x) { return std::isnan(x); }
inline bool isinf(float x) { return std::isinf(x); }

// ----------------------------------------------------------------------------------------------
// componentwise vector intrinsics (float2/3/4)
#define HL_UN(N, f) inline vec<float, N> f(const vec<float, N>& a) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = f(a.d[i]); return r; }
#define HL_BIN(N, f)                                                                                                                                        \
    inline vec<float, N> f(const vec<float, N>& a, const vec<float, N>& b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = f(a.d[i], b.d[i]); return r; } \
    inline vec<float, N> f(const vec<float, N>& a, float b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = f(a.d[i], b); return r; }                     \
    inline vec<float, N> f(float a, const vec<float, N>& b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = f(a, b.d[i]); return r; }
#define HL_FLOATVEC(N)                                                                                                                  \
    HL_UN(N, abs) HL_UN(N, floor) HL_UN(N, ceil) HL_UN(N, round) HL_UN(N, trunc) HL_UN(N, frac) HL_UN(N, saturate) HL_UN(N, sqrt)        \
    HL_UN(N, rsqrt) HL_UN(N, exp) HL_UN(N, exp2) HL_UN(N, log) HL_UN(N, log2) HL_UN(N, log10) HL_UN(N, sin) HL_UN(N, cos) HL_UN(N, tan)  \
    HL_UN(N, asin) HL_UN(N, acos) HL_UN(N, atan) HL_UN(N, rcp) HL_UN(N, sign)                                                            \
    HL_BIN(N, min) HL_BIN(N, max) HL_BIN(N, pow) HL_BIN(N, step) HL_BIN(N, atan2) HL_BIN(N, fmod)                                        \
    inline vec<float, N> clamp(const vec<float, N>& x, const vec<float, N>& a, const vec<float, N>& b) { return min(max(x, a), b); }     \
    inline vec<float, N> clamp(const vec<float, N>& x, float a, float b) { return min(max(x, a), b); }                                   \
    inline vec<float, N> lerp(const vec<float, N>& a, const vec<float, N>& b, const vec<float, N>& t) { return a + t * (b - a); }        \
    inline vec<float, N> lerp(const vec<float, N>& a, const vec<float, N>& b, float t) { return a + t * (b - a); }                       \
    inline vec<float, N> smoothstep(const vec<float, N>& a, const vec<float, N>& b, const vec<float, N>& x)                              \
    { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = smoothstep(a.d[i], b.d[i], x.d[i]); return r; }                             \
    inline float dot(const vec<float, N>& a, const vec<float, N>& b) { float s = a.d[0] * b.d[0]; for (int i = 1; i < N; ++i) s += a.d[i] * b.d[i]; return s; } \
    inline float length(const vec<float, N>& a) { return std::sqrt(dot(a, a)); }                                                         \
    inline float distance(const vec<float, N>& a, const vec<float, N>& b) { return length(a - b); }                                      \
    inline vec<float, N> normalize(const vec<float, N>& a) { return a * (1.0f / std::sqrt(dot(a, a))); }                                  \
    inline vec<float, N> reflect(const vec<float, N>& i, const vec<float, N>& n) { return i - 2.0f * dot(n, i) * n; }                    \
    inline vec<float, N> GreaterEqual(const vec<float, N>& a, const vec<float, N>& b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] >= b.d[i] ? 1.f : 0.f; return r; } \
    inline vec<float, N> LessEqual(const vec<float, N>& a, const vec<float, N>& b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] <= b.d[i] ? 1.f : 0.f; return r; }    \
    inline vec<float, N> Less(const vec<float, N>& a, const vec<float, N>& b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] < b.d[i] ? 1.f : 0.f; return r; }          \
    inline vec<float, N> Greater(const vec<float, N>& a, const vec<float, N>& b) { vec<float, N> r; for (int i = 0; i < N; ++i) r.d[i] = a.d[i] > b.d[i] ? 1.f : 0.f; return r; }
HL_FLOATVEC(2)
HL_FLOATVEC(3)
HL_FLOATVEC(4)
inline float3 cross(const float3& a, const float3& b) { return float3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

#define HL_INTVEC(T, N)                                                                                                                           \
    inline vec<T, N> min(const vec<T, N>& a, const vec<T, N>& b) { vec<T, N> r; for (int i = 0; i < N; ++i) r.d[i] = min(a.d[i], b.d[i]); return r; } \
    inline vec<T, N> max(const vec<T, N>& a, const vec<T, N>& b) { vec<T, N> r; for (int i = 0; i < N; ++i) r.d[i] = max(a.d[i], b.d[i]); return r; } \
    inline vec<T, N> clamp(const vec<T, N>& x, const vec<T, N>& a, const vec<T, N>& b) { return min(max(x, a), b); }
HL_FORALL_N(HL_INTVEC, int)
HL_FORALL_N(HL_INTVEC, uint)

// ----------------------------------------------------------------------------------------------
// matrices: row-major storage, row-vector convention (mul(v, M) = sum_i v[i] * row_i)
struct float3x3
{
    float3 r[3];
    float3& operator[](int i) { return r[i]; }
    const float3& operator[](int i) const { return r[i]; }
};
struct float4x4
{
    float4 r[4];
    float4& operator[](int i) { return r[i]; }
    const float4& operator[](int i) const { return r[i]; }
};
static_assert(sizeof(float4x4) == 64, "matrix layout");
inline float3x3 MatrixFromRows(const float3& a, const float3& b, const float3& c) { float3x3 m; m.r[0] = a; m.r[1] = b; m.r[2] = c; return m; }
inline float4x4 MatrixFromRows(const float4& a, const float4& b, const float4& c, const float4& d) { float4x4 m; m.r[0] = a; m.r[1] = b; m.r[2] = c; m.r[3] = d; return m; }
inline float3 mul(const float3& v, const float3x3& m) { return v.x * m.r[0] + v.y * m.r[1] + v.z * m.r[2]; }
inline float3 mul(const float3x3& m, const float3& v) { return float3(dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v)); }
inline float4 mul(const float4& v, const float4x4& m) { return v.x * m.r[0] + v.y * m.r[1] + v.z * m.r[2] + v.w * m.r[3]; }
inline float4 mul(const float4x4& m, const float4& v) { return float4(dot(m.r[0], v), dot(m.r[1], v), dot(m.r[2], v), dot(m.r[3], v)); }

// ----------------------------------------------------------------------------------------------
// DiligentCore HLSL platform definitions, D3D/Vulkan path
#define NDC_MIN_Z 0.0
#define MATRIX_ELEMENT(m, r, c) ((m)[r][c])
#define F3NDC_XYZ_TO_UVD_SCALE float3(0.5, -0.5, 1.0)
inline float2 NormalizedDeviceXYToTexUV(const float2& xy) { return float2(0.5, 0.5) + float2(0.5, -0.5) * xy; }
inline float2 TexUVToNormalizedDeviceXY(const float2& uv) { return (uv - float2(0.5, 0.5)) * float2(2.0, -2.0); }
inline float  NormalizedDeviceZToDepth(float z) { return z; }
inline float  DepthToNormalizedDeviceZ(float d) { return d; }

// ----------------------------------------------------------------------------------------------
// execution context: discard flag + quad derivatives (two-phase 2x2 emulation)
struct ExecCtx
{
    bool  discarded = false;
    int   quad_phase = -1; // -1: derivatives unavailable (return 0); 0: record; 1: replay
    int   quad_lane  = 0;  // 0..3 = (x&1) + 2*(y&1)
    int   call_idx   = 0;
    float rec[4][8];
};
inline thread_local ExecCtx g_ctx;
#define discard do { ::hlsl::g_ctx.discarded = true; return {}; } while (0)

inline float hl_deriv(float v, bool is_y)
{
    ExecCtx& c = g_ctx;
    if (c.quad_phase < 0) return 0.0f;
    int k = c.call_idx++;
    if (c.quad_phase == 0) { c.rec[c.quad_lane][k] = v; return 0.0f; }
    // fine derivatives inside the 2x2 quad: ddx = right - left of this row, ddy = bottom - top of this column
    int l = c.quad_lane;
    if (!is_y) { int row = l & 2; return c.rec[row + 1][k] - c.rec[row][k]; }
    int col = l & 1;
    return c.rec[col + 2][k] - c.rec[col][k];
}
inline float ddx(float v) { return hl_deriv(v, false); }
inline float ddy(float v) { return hl_deriv(v, true); }
// vector derivatives are only referenced by out-of-scope material code (normal mapping); never executed here.
inline float2 ddx(const float2&) { return float2(0.f, 0.f); }
inline float2 ddy(const float2&) { return float2(0.f, 0.f); }
inline float3 ddx(const float3&) { return float3(0.f, 0.f, 0.f); }
inline float3 ddy(const float3&) { return float3(0.f, 0.f, 0.f); }

// ----------------------------------------------------------------------------------------------
// textures
struct Image
{
    const float* data = nullptr;
    int w = 0, h = 0, c = 0;
};
enum { ADDR_CLAMP = 0, ADDR_BORDER = 1, ADDR_WRAP = 2 };
struct SamplerState
{
    bool linear     = false;
    bool mip_linear = false;
    int  addr       = ADDR_CLAMP;
};
inline const SamplerState Sam_PointClamp{false, false, ADDR_CLAMP};
inline const SamplerState Sam_LinearClamp{true, true, ADDR_CLAMP};
inline const SamplerState Sam_PointWrap{false, false, ADDR_WRAP};
inline const SamplerState Sam_LinearBorder{true, false, ADDR_BORDER};

struct TexStorage
{
    Image mip[16];
    int   mips = 0;
};

inline float4 hl_fetch(const Image& im, int x, int y)
{
    float4 r(0.f, 0.f, 0.f, 0.f);
    if (x < 0 || y < 0 || x >= im.w || y >= im.h || !im.data) return r; // D3D: out-of-bounds Load returns 0
    const float* p = im.data + (size_t(y) * im.w + x) * im.c;
    for (int k = 0; k < im.c && k < 4; ++k) r.d[k] = p[k];
    return r;
}
inline int hl_addr(int v, int n, int mode, bool& oob)
{
    if (mode == ADDR_WRAP) { v %= n; if (v < 0) v += n; return v; }
    if (mode == ADDR_BORDER) { if (v < 0 || v >= n) oob = true; return v; }
    return v < 0 ? 0 : (v >= n ? n - 1 : v);
}
inline float4 hl_sample_level(const Image& im, const SamplerState& s, float u, float v)
{
    if (!s.linear)
    {
        bool oob = false;
        int x = hl_addr(int(std::floor(u * float(im.w))), im.w, s.addr, oob);
        int y = hl_addr(int(std::floor(v * float(im.h))), im.h, s.addr, oob);
        return oob ? float4(0.f, 0.f, 0.f, 0.f) : hl_fetch(im, x, y);
    }
    float fx = u * float(im.w) - 0.5f, fy = v * float(im.h) - 0.5f;
    float x0f = std::floor(fx), y0f = std::floor(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    float4 acc(0.f, 0.f, 0.f, 0.f);
    const float wgt[4] = {(1.f - wx) * (1.f - wy), wx * (1.f - wy), (1.f - wx) * wy, wx * wy};
    for (int t = 0; t < 4; ++t)
    {
        bool oob = false;
        int x = hl_addr(x0 + (t & 1), im.w, s.addr, oob);
        int y = hl_addr(y0 + (t >> 1), im.h, s.addr, oob);
        if (!oob) acc += hl_fetch(im, x, y) * wgt[t];
    }
    return acc;
}
inline float4 hl_sample(const TexStorage& t, const SamplerState& s, float u, float v, float lod)
{
    float maxl = float(t.mips - 1);
    lod = lod < 0.f ? 0.f : (lod > maxl ? maxl : lod);
    if (!s.mip_linear)
    {
        int l = int(std::floor(lod + 0.5f)); // nearest mip (D3D/Vulkan point-mip rule)
        if (l > t.mips - 1) l = t.mips - 1;
        return hl_sample_level(t.mip[l], s, u, v);
    }
    int   l0 = int(std::floor(lod));
    int   l1 = l0 + 1 < t.mips ? l0 + 1 : l0;
    float f  = lod - float(l0);
    float4 a = hl_sample_level(t.mip[l0], s, u, v);
    if (f == 0.f || l1 == l0) return a;
    float4 b = hl_sample_level(t.mip[l1], s, u, v);
    return a + (b - a) * f;
}

template <class T> struct hl_cvt;
template <> struct hl_cvt<float>  { static float  get(const float4& v) { return v.x; } };
template <> struct hl_cvt<float2> { static float2 get(const float4& v) { return float2(v.x, v.y); } };
template <> struct hl_cvt<float3> { static float3 get(const float4& v) { return float3(v.x, v.y, v.z); } };
template <> struct hl_cvt<float4> { static float4 get(const float4& v) { return v; } };
template <> struct hl_cvt<uint>   { static uint   get(const float4& v) { return uint(v.x); } };

template <class T = float4> struct Texture2D_
{
    TexStorage s;
    template <class C> T Load(const C& c) const
    {
        vec<int, 3> p(c);
        int l = p.z;
        if (l < 0 || l >= s.mips) return hl_cvt<T>::get(float4(0.f, 0.f, 0.f, 0.f));
        return hl_cvt<T>::get(hl_fetch(s.mip[l], p.x, p.y));
    }
    T SampleLevel(const SamplerState& sm, const float2& uv, float lod) const { return hl_cvt<T>::get(hl_sample(s, sm, uv.x, uv.y, lod)); }
    T Sample(const SamplerState& sm, const float2& uv) const { return SampleLevel(sm, uv, 0.0f); } // only used on 1-mip LUTs
    template <class A> void GetDimensions(A& w, A& h) const { w = A(s.mip[0].w); h = A(s.mip[0].h); }
    template <class A, class B> void GetDimensions(int mip, A& w, A& h, B& n) const { w = A(s.mip[mip].w); h = A(s.mip[mip].h); n = B(s.mips); }
};

// Cube map: 6 faces per mip, D3D face order (+X,-X,+Y,-Y,+Z,-Z), each face w*w texels, faces contiguous.
// Bilinear taps that fall outside the face are resolved by re-projecting the (extended-plane) tap position
// onto the cube and fetching the nearest texel of the face it lands on ("seamless by re-projection"); this is
// the filtering contract shared by oracle and HIP (IBL sampling).
struct CubeStorage
{
    Image mip[16]; // h = 6*w
    int   mips = 0;
};
inline void hl_cube_face_uv(const float3& d, int& face, float& u, float& v)
{
    float ax = std::fabs(d.x), ay = std::fabs(d.y), az = std::fabs(d.z);
    float ma, sc, tc;
    if (ax >= ay && ax >= az) { ma = ax; if (d.x >= 0) { face = 0; sc = -d.z; tc = -d.y; } else { face = 1; sc = d.z; tc = -d.y; } }
    else if (ay >= az)        { ma = ay; if (d.y >= 0) { face = 2; sc = d.x; tc = d.z; }  else { face = 3; sc = d.x; tc = -d.z; } }
    else                      { ma = az; if (d.z >= 0) { face = 4; sc = d.x; tc = -d.y; } else { face = 5; sc = -d.x; tc = -d.y; } }
    u = 0.5f * (sc / ma + 1.0f);
    v = 0.5f * (tc / ma + 1.0f);
}
inline float3 hl_cube_dir(int face, float u, float v) // u,v in [0,1] (may exceed for extended plane)
{
    float sc = 2.0f * u - 1.0f, tc = 2.0f * v - 1.0f;
    switch (face)
    {
        case 0: return float3(1.f, -tc, -sc);
        case 1: return float3(-1.f, -tc, sc);
        case 2: return float3(sc, 1.f, tc);
        case 3: return float3(sc, -1.f, -tc);
        case 4: return float3(sc, -tc, 1.f);
        default: return float3(-sc, -tc, -1.f);
    }
}
inline float4 hl_cube_texel(const Image& im, int face, int x, int y)
{
    int n = im.w;
    if (x < 0 || y < 0 || x >= n || y >= n)
    {
        float3 d = hl_cube_dir(face, (float(x) + 0.5f) / float(n), (float(y) + 0.5f) / float(n));
        float u, v;
        hl_cube_face_uv(d, face, u, v);
        x = int(std::floor(u * float(n)));
        y = int(std::floor(v * float(n)));
        x = x < 0 ? 0 : (x >= n ? n - 1 : x);
        y = y < 0 ? 0 : (y >= n ? n - 1 : y);
    }
    return hl_fetch(im, x, face * n + y);
}
inline float4 hl_cube_sample_level(const Image& im, const float3& dir)
{
    int face; float u, v;
    hl_cube_face_uv(dir, face, u, v);
    int n = im.w;
    float fx = u * float(n) - 0.5f, fy = v * float(n) - 0.5f;
    float x0f = std::floor(fx), y0f = std::floor(fy);
    float wx = fx - x0f, wy = fy - y0f;
    int x0 = int(x0f), y0 = int(y0f);
    float4 acc = hl_cube_texel(im, face, x0, y0) * ((1.f - wx) * (1.f - wy));
    acc += hl_cube_texel(im, face, x0 + 1, y0) * (wx * (1.f - wy));
    acc += hl_cube_texel(im, face, x0, y0 + 1) * ((1.f - wx) * wy);
    acc += hl_cube_texel(im, face, x0 + 1, y0 + 1) * (wx * wy);
    return acc;
}
// Texture2DArray<float> + SamplerComparisonState for the shadow map of the punctual lights (PCF.fxh).  The one comparison sampler the reference uses is
// Sam_ComparisonLinearClamp of DiligentCore's CommonlyUsedStates.h (un-vendored; recalled: linear min / mag / mip, CLAMP, COMPARISON_FUNC_LESS): SampleCmpLevelZero
// compares the reference value with each of the four texels of the bilinear footprint ("reference < texel" -> 1) and blends the results with the bilinear weights;
// the array slice is the coordinate rounded to the nearest integer and clamped.
struct SamplerComparisonState
{
    bool less = true;
};
inline const SamplerComparisonState Sam_ComparisonLinearClamp{true};
template <class T = float> struct Texture2DArray_
{
    Image slice[32];
    int   slices = 0;
    template <class A, class B> void GetDimensions(A& w, A& h, B& n) const { w = A(slice[0].w); h = A(slice[0].h); n = B(slices); }
    float SampleCmpLevelZero(const SamplerComparisonState&, const float3& uvs, float ref) const
    {
        int s = int(std::floor(uvs.z + 0.5f));
        s = s < 0 ? 0 : (s > slices - 1 ? slices - 1 : s);
        const Image& im = slice[s];
        float fx = uvs.x * float(im.w) - 0.5f, fy = uvs.y * float(im.h) - 0.5f;
        float x0f = std::floor(fx), y0f = std::floor(fy);
        float wx = fx - x0f, wy = fy - y0f;
        int x0 = int(x0f), y0 = int(y0f);
        const float wgt[4] = {(1.f - wx) * (1.f - wy), wx * (1.f - wy), (1.f - wx) * wy, wx * wy};
        float acc = 0.0f;
        for (int t = 0; t < 4; ++t)
        {
            bool oob = false;
            int x = hl_addr(x0 + (t & 1), im.w, ADDR_CLAMP, oob), y = hl_addr(y0 + (t >> 1), im.h, ADDR_CLAMP, oob);
            acc += (ref < hl_fetch(im, x, y).x ? 1.0f : 0.0f) * wgt[t];
        }
        return acc;
    }
};
inline void ref_bind_array(Texture2DArray_<float>& t, const float* const* data, int n, int w, int h)
{
    t.slices = n;
    for (int i = 0; i < n; ++i) t.slice[i] = Image{data[i], w, h, 1};
}

struct TextureCube
{
    CubeStorage s;
    float4 SampleLevel(const SamplerState&, const float3& dir, float lod) const
    {
        float maxl = float(s.mips - 1);
        lod = lod < 0.f ? 0.f : (lod > maxl ? maxl : lod);
        int   l0 = int(std::floor(lod));
        int   l1 = l0 + 1 < s.mips ? l0 + 1 : l0;
        float f  = lod - float(l0);
        float4 a = hl_cube_sample_level(s.mip[l0], dir);
        if (f == 0.f || l1 == l0) return a;
        float4 b = hl_cube_sample_level(s.mip[l1], dir);
        return a + (b - a) * f;
    }
    float4 Sample(const SamplerState& sm, const float3& dir) const { return SampleLevel(sm, dir, 0.0f); }
};

// HLSL storage-class / parameter keywords that have no C++ meaning here ('in', 'out', 'inout' are rewritten by ref_prep.py)
#define cbuffer inline namespace

} // namespace hlsl
