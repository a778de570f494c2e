// dg_mock.hpp -- TEST INFRASTRUCTURE ONLY (oracle/refhost: the reference's HOST code compiled where it lies, never part of the product).
//
// A recording stand-in for the part of DiligentCore that DiligentFX's post-process classes drive (PostFXContext, ScreenSpaceAmbientOcclusion, ScreenSpaceReflection,
// TemporalAntiAliasing, Bloom and PostFXRenderTechnique, under /root/reference/PostProcess).  DiligentCore itself is not in the reference tree, so those sources cannot
// be built as shipped; against this header they compile unmodified, and every call they make on the device context -- SetRenderTargets, SetPipelineState,
// CommitShaderResources, Draw, ClearRenderTarget, CopyTexture, UpdateBuffer / MapBuffer, debug groups -- is written to a command list with every operand resolved:
// the pipeline's name and pixel-shader entry point + macros, the render targets as (texture, mip), every bound shader variable by NAME as (texture, first mip, mip
// count) or buffer, instance counts and start vertices, clear colours, buffer bytes.  oracle/refhost.py replays such a list on numpy planes by calling the matching
// pass of oracle/_ref (the reference's shader source compiled for the CPU): reference host code + reference shaders = the reference's frame, which pins the pass order,
// clears, ping-pong, reset rules and mip loops that oracle/cpu_chain.py and csrc/api_*.cpp restate (tests/test_host_sequence_vs_ref.py).
//
// Semantics kept from DiligentCore where the host code depends on them: reference counting (RefCntAutoPtr / Release), static vs mutable / dynamic shader variables
// (static ones live in the pipeline and are copied into a binding created with InitStaticResources), default texture views, ResourceRegistry, ShaderMacroHelper,
// Timer (driven by the harness), pipeline status (always ready).  Everything else is the smallest thing that lets the sources compile.  The names are DiligentCore's
// public API names by necessity; no DiligentCore source was available or used.
#pragma once
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#ifndef _countof
#    define _countof(a) (sizeof(a) / sizeof((a)[0]))
#endif
#define DILIGENT_CONSTEXPR constexpr

namespace Diligent
{
using Uint8  = uint8_t;
using Uint16 = uint16_t;
using Uint32 = uint32_t;
using Uint64 = uint64_t;
using Int8   = int8_t;
using Int16  = int16_t;
using Int32  = int32_t;
using Int64  = int64_t;
using Float32 = float;
using Char    = char;
using Bool    = bool;
using uint    = uint32_t; // the shader structure headers are included inside `namespace HLSL` nested in Diligent
static constexpr Bool True  = true;
static constexpr Bool False = false;

// ------------------------------------------------------------------------------------------------ recorder
struct Recorder
{
    std::vector<std::string> lines;
    std::vector<std::string> groups; // open debug groups
    std::vector<std::string> errors;
    float                    timerElapsed = 0.0f; // what every Timer::GetElapsedTimef() returns (the harness sets it: AlphaInterpolation)
    int                      nextId       = 1;
    static Recorder& Get()
    {
        static Recorder r;
        return r;
    }
    void Emit(const std::string& s) { lines.push_back(s); }
    void Error(const std::string& s)
    {
        errors.push_back(s);
        lines.push_back("{\"op\":\"error\",\"what\":\"" + s + "\"}");
    }
};
inline std::string JsonStr(const char* s)
{
    std::string o = "\"";
    for (const char* p = s ? s : ""; *p; ++p)
    {
        if (*p == '"' || *p == '\\') o += '\\';
        if (*p == '\n') { o += "\\n"; continue; }
        o += *p;
    }
    return o + "\"";
}
inline std::string Base64(const void* data, size_t n)
{
    static const char* T = "ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+/";
    const unsigned char* p = static_cast<const unsigned char*>(data);
    std::string o;
    o.reserve((n + 2) / 3 * 4);
    for (size_t i = 0; i < n; i += 3)
    {
        const unsigned v = (p[i] << 16) | ((i + 1 < n ? p[i + 1] : 0) << 8) | (i + 2 < n ? p[i + 2] : 0);
        o += T[(v >> 18) & 63];
        o += T[(v >> 12) & 63];
        o += i + 1 < n ? T[(v >> 6) & 63] : '=';
        o += i + 2 < n ? T[v & 63] : '=';
    }
    return o;
}

// ------------------------------------------------------------------------------------------------ logging / checks
#define DEV_CHECK_ERR(cond, ...)                                                              \
    do {                                                                                      \
        if (!(cond)) ::Diligent::Recorder::Get().Error(std::string("DEV_CHECK_ERR: ") + #cond); \
    } while (0)
#define VERIFY_EXPR(cond) DEV_CHECK_ERR(cond, "")
#define VERIFY(cond, ...) DEV_CHECK_ERR(cond, "")
#define DEV_ERROR(...) ::Diligent::Recorder::Get().Error("DEV_ERROR")
#define LOG_ERROR_MESSAGE(...) ::Diligent::Recorder::Get().Error("LOG_ERROR_MESSAGE")
#define LOG_WARNING_MESSAGE(...) \
    do {                         \
    } while (0)
#define UNEXPECTED(...) ::Diligent::Recorder::Get().Error("UNEXPECTED")

#define DEFINE_FLAG_ENUM_OPERATORS(ENUMTYPE)                                                                                                                          \
    inline constexpr ENUMTYPE  operator|(ENUMTYPE a, ENUMTYPE b) { return static_cast<ENUMTYPE>(static_cast<std::underlying_type_t<ENUMTYPE>>(a) | static_cast<std::underlying_type_t<ENUMTYPE>>(b)); } \
    inline constexpr ENUMTYPE  operator&(ENUMTYPE a, ENUMTYPE b) { return static_cast<ENUMTYPE>(static_cast<std::underlying_type_t<ENUMTYPE>>(a) & static_cast<std::underlying_type_t<ENUMTYPE>>(b)); } \
    inline constexpr ENUMTYPE  operator^(ENUMTYPE a, ENUMTYPE b) { return static_cast<ENUMTYPE>(static_cast<std::underlying_type_t<ENUMTYPE>>(a) ^ static_cast<std::underlying_type_t<ENUMTYPE>>(b)); } \
    inline constexpr ENUMTYPE  operator~(ENUMTYPE a) { return static_cast<ENUMTYPE>(~static_cast<std::underlying_type_t<ENUMTYPE>>(a)); }                              \
    inline ENUMTYPE&           operator|=(ENUMTYPE& a, ENUMTYPE b) { return a = a | b; }                                                                               \
    inline ENUMTYPE&           operator&=(ENUMTYPE& a, ENUMTYPE b) { return a = a & b; }                                                                               \
    inline ENUMTYPE&           operator^=(ENUMTYPE& a, ENUMTYPE b) { return a = a ^ b; }

// ------------------------------------------------------------------------------------------------ BasicMath
template <class T> struct Vector2
{
    T x{}, y{};
    constexpr Vector2() = default;
    constexpr Vector2(T _x, T _y) : x{_x}, y{_y} {}
    constexpr Vector2 operator*(const Vector2& r) const { return {x * r.x, y * r.y}; }
    constexpr Vector2 operator+(const Vector2& r) const { return {x + r.x, y + r.y}; }
    constexpr Vector2 operator-(const Vector2& r) const { return {x - r.x, y - r.y}; }
    constexpr Vector2 operator*(T s) const { return {x * s, y * s}; }
    constexpr bool    operator==(const Vector2& r) const { return x == r.x && y == r.y; }
    constexpr bool    operator!=(const Vector2& r) const { return !(*this == r); }
    T&       operator[](size_t i) { return (&x)[i]; }
    const T& operator[](size_t i) const { return (&x)[i]; }
};
template <class T> constexpr Vector2<T> operator*(T s, const Vector2<T>& v) { return {s * v.x, s * v.y}; }
static constexpr float PI_F = 3.14159265358979323846f; // BasicMath.hpp
template <class T> struct Vector3
{
    T x{}, y{}, z{};
    constexpr Vector3() = default;
    constexpr Vector3(T _x, T _y, T _z) : x{_x}, y{_y}, z{_z} {}
    constexpr Vector3 operator*(T s) const { return {x * s, y * s, z * s}; }
    constexpr Vector3 operator+(const Vector3& r) const { return {x + r.x, y + r.y, z + r.z}; }
    constexpr bool    operator==(const Vector3& r) const { return x == r.x && y == r.y && z == r.z; }
    T&       operator[](size_t i) { return (&x)[i]; }
    const T& operator[](size_t i) const { return (&x)[i]; }
};
template <class T> struct Vector4
{
    T x{}, y{}, z{}, w{};
    constexpr Vector4() = default;
    constexpr Vector4(T _x, T _y, T _z, T _w) : x{_x}, y{_y}, z{_z}, w{_w} {}
    constexpr Vector4(const Vector2<T>& xy, T _z, T _w) : x{xy.x}, y{xy.y}, z{_z}, w{_w} {}
    constexpr Vector4(const Vector3<T>& v, T _w) : x{v.x}, y{v.y}, z{v.z}, w{_w} {}
    constexpr bool operator==(const Vector4& r) const { return x == r.x && y == r.y && z == r.z && w == r.w; }
    constexpr bool operator!=(const Vector4& r) const { return !(*this == r); }
    T*       Data() { return &x; }
    const T* Data() const { return &x; }
    T&       operator[](size_t i) { return (&x)[i]; }
    const T& operator[](size_t i) const { return (&x)[i]; }
};
template <class T> struct Matrix4x4
{
    union
    {
        struct
        {
            T _11, _12, _13, _14, _21, _22, _23, _24, _31, _32, _33, _34, _41, _42, _43, _44;
        };
        struct
        {
            T m00, m01, m02, m03, m10, m11, m12, m13, m20, m21, m22, m23, m30, m31, m32, m33;
        };
        T m[4][4];
    };
    constexpr Matrix4x4() : m{} {}
    T*       operator[](size_t r) { return m[r]; }
    const T* operator[](size_t r) const { return m[r]; }
    T*       Data() { return &m[0][0]; }
    const T* Data() const { return &m[0][0]; }
    static Matrix4x4 Identity()
    {
        Matrix4x4 r;
        r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = T(1);
        return r;
    }
};
template <class T> struct Matrix3x3
{
    T m[3][3] = {};
};
using float2   = Vector2<float>;
using float3   = Vector3<float>;
using float4   = Vector4<float>;
using int2     = Vector2<Int32>;
using int3     = Vector3<Int32>;
using int4     = Vector4<Int32>;
using uint2    = Vector2<Uint32>;
using uint3    = Vector3<Uint32>;
using uint4    = Vector4<Uint32>;
using float4x4 = Matrix4x4<float>;
using float3x3 = Matrix3x3<float>;

template <typename T> inline void HashCombine(size_t& seed, const T& v) { seed ^= std::hash<T>{}(v) + 0x9e3779b9 + (seed << 6) + (seed >> 2); }
inline void                        HashCombineAll(size_t&) {}
template <typename T, typename... R> inline void HashCombineAll(size_t& seed, const T& v, const R&... r)
{
    if constexpr (std::is_enum_v<T>) HashCombine(seed, static_cast<std::underlying_type_t<T>>(v));
    else HashCombine(seed, v);
    HashCombineAll(seed, r...);
}
template <typename... A> inline size_t ComputeHash(const A&... a)
{
    size_t seed = 0;
    HashCombineAll(seed, a...);
    return seed;
}

// ------------------------------------------------------------------------------------------------ Timer
class Timer
{
public:
    void   Restart() { ++restarts; }
    float  GetElapsedTimef() const { return Recorder::Get().timerElapsed; }
    double GetElapsedTime() const { return Recorder::Get().timerElapsed; }
    int    restarts = 0;
};

// ------------------------------------------------------------------------------------------------ reference counting
struct INTERFACE_ID
{
    int id;
};
struct IObject
{
    virtual ~IObject() = default;
    void AddRef() { ++refs; }
    void Release()
    {
        if (--refs == 0) delete this;
    }
    int refs = 0;
};
template <class T> class RefCntAutoPtr
{
public:
    RefCntAutoPtr() = default;
    RefCntAutoPtr(std::nullptr_t) {}
    explicit RefCntAutoPtr(T* p) : m_p{p}
    {
        if (m_p) m_p->AddRef();
    }
    template <class U> RefCntAutoPtr(U* p, const INTERFACE_ID&) : m_p{dynamic_cast<T*>(p)}
    {
        if (m_p) m_p->AddRef();
    }
    RefCntAutoPtr(const RefCntAutoPtr& o) : m_p{o.m_p}
    {
        if (m_p) m_p->AddRef();
    }
    template <class U, class = std::enable_if_t<std::is_convertible_v<U*, T*>>> RefCntAutoPtr(const RefCntAutoPtr<U>& o) : m_p{o.RawPtr()}
    {
        if (m_p) m_p->AddRef();
    }
    RefCntAutoPtr(RefCntAutoPtr&& o) noexcept : m_p{o.m_p} { o.m_p = nullptr; }
    ~RefCntAutoPtr() { Release(); }
    RefCntAutoPtr& operator=(const RefCntAutoPtr& o)
    {
        if (o.m_p) o.m_p->AddRef();
        Release();
        m_p = o.m_p;
        return *this;
    }
    RefCntAutoPtr& operator=(RefCntAutoPtr&& o) noexcept
    {
        if (this != std::addressof(o))
        {
            Release();
            m_p   = o.m_p;
            o.m_p = nullptr;
        }
        return *this;
    }
    RefCntAutoPtr& operator=(T* p)
    {
        if (p) p->AddRef();
        Release();
        m_p = p;
        return *this;
    }
    void Release()
    {
        if (m_p)
        {
            T* p = m_p;
            m_p  = nullptr;
            p->Release();
        }
    }
    void Attach(T* p)
    {
        Release();
        m_p = p;
    }
    T* Detach()
    {
        T* p = m_p;
        m_p  = nullptr;
        return p;
    }
    T*   RawPtr() const { return m_p; }
    template <class U> U* RawPtr() const { return static_cast<U*>(m_p); }
    T*   operator->() const { return m_p; }
    T&   operator*() const { return *m_p; }
    operator T*() const { return m_p; }
    explicit operator bool() const { return m_p != nullptr; }
    bool operator!() const { return m_p == nullptr; }
    // `&ptr` as an out-parameter: the pointee written through the returned T** is adopted on destruction of the helper
    class DoublePtrHelper
    {
    public:
        explicit DoublePtrHelper(RefCntAutoPtr& o) : m_o{o}, m_raw{o.m_p} {}
        ~DoublePtrHelper()
        {
            if (m_raw != m_o.m_p) m_o.Attach(m_raw);
        }
        operator T**() { return &m_raw; }
        template <class U, class = std::enable_if_t<std::is_base_of_v<U, T>>> operator U**() { return reinterpret_cast<U**>(&m_raw); }
        T*& operator*() { return m_raw; }

    private:
        RefCntAutoPtr& m_o;
        T*             m_raw;
    };
    DoublePtrHelper operator&() { return DoublePtrHelper{*this}; }

private:
    T* m_p = nullptr;
};

// ------------------------------------------------------------------------------------------------ enums of the graphics API (the values the host code uses)
enum RENDER_DEVICE_TYPE : Uint32 { RENDER_DEVICE_TYPE_UNDEFINED = 0, RENDER_DEVICE_TYPE_D3D11, RENDER_DEVICE_TYPE_D3D12, RENDER_DEVICE_TYPE_GL, RENDER_DEVICE_TYPE_GLES, RENDER_DEVICE_TYPE_VULKAN, RENDER_DEVICE_TYPE_METAL, RENDER_DEVICE_TYPE_WEBGPU };
enum TEXTURE_FORMAT : Uint16
{
    TEX_FORMAT_UNKNOWN = 0, TEX_FORMAT_RGBA32_FLOAT, TEX_FORMAT_RGBA16_FLOAT, TEX_FORMAT_RG16_FLOAT, TEX_FORMAT_R16_FLOAT, TEX_FORMAT_R16_UNORM, TEX_FORMAT_R32_FLOAT, TEX_FORMAT_R8_UNORM, TEX_FORMAT_R8_UINT,
    TEX_FORMAT_RG8_UNORM, TEX_FORMAT_R11G11B10_FLOAT, TEX_FORMAT_D16_UNORM, TEX_FORMAT_D32_FLOAT, TEX_FORMAT_RGBA8_UNORM, TEX_FORMAT_RGBA8_UNORM_SRGB, TEX_FORMAT_RG32_FLOAT, TEX_FORMAT_D24_UNORM_S8_UINT,
    TEX_FORMAT_D32_FLOAT_S8X24_UINT, TEX_FORMAT_R32_UINT, TEX_FORMAT_NUM_FORMATS
};
inline size_t TexelBytes(TEXTURE_FORMAT f) // of the formats the host code uploads initial data in
{
    return f == TEX_FORMAT_R8_UINT || f == TEX_FORMAT_R8_UNORM ? 1 : f == TEX_FORMAT_RG32_FLOAT ? 8 : f == TEX_FORMAT_RGBA32_FLOAT ? 16 : 4;
}
inline const char* FormatName(TEXTURE_FORMAT f)
{
    static const char* N[] = {"UNKNOWN", "RGBA32_FLOAT", "RGBA16_FLOAT", "RG16_FLOAT", "R16_FLOAT", "R16_UNORM", "R32_FLOAT", "R8_UNORM", "R8_UINT", "RG8_UNORM", "R11G11B10_FLOAT", "D16_UNORM", "D32_FLOAT",
                              "RGBA8_UNORM", "RGBA8_UNORM_SRGB", "RG32_FLOAT", "D24_UNORM_S8_UINT", "D32_FLOAT_S8X24_UINT", "R32_UINT"};
    return f < TEX_FORMAT_NUM_FORMATS ? N[f] : "?";
}
enum RESOURCE_DIMENSION : Uint8 { RESOURCE_DIM_UNDEFINED = 0, RESOURCE_DIM_BUFFER, RESOURCE_DIM_TEX_1D, RESOURCE_DIM_TEX_1D_ARRAY, RESOURCE_DIM_TEX_2D, RESOURCE_DIM_TEX_2D_ARRAY, RESOURCE_DIM_TEX_3D, RESOURCE_DIM_TEX_CUBE, RESOURCE_DIM_TEX_CUBE_ARRAY };
enum BIND_FLAGS : Uint32 { BIND_NONE = 0, BIND_VERTEX_BUFFER = 1, BIND_INDEX_BUFFER = 2, BIND_UNIFORM_BUFFER = 4, BIND_SHADER_RESOURCE = 8, BIND_STREAM_OUTPUT = 16, BIND_RENDER_TARGET = 32, BIND_DEPTH_STENCIL = 64, BIND_UNORDERED_ACCESS = 128 };
DEFINE_FLAG_ENUM_OPERATORS(BIND_FLAGS)
enum USAGE : Uint8 { USAGE_IMMUTABLE = 0, USAGE_DEFAULT, USAGE_DYNAMIC, USAGE_STAGING };
enum CPU_ACCESS_FLAGS : Uint8 { CPU_ACCESS_NONE = 0, CPU_ACCESS_READ = 1, CPU_ACCESS_WRITE = 2 };
DEFINE_FLAG_ENUM_OPERATORS(CPU_ACCESS_FLAGS)
enum MAP_TYPE : Uint8 { MAP_READ = 1, MAP_WRITE = 2, MAP_READ_WRITE = 3 };
enum MAP_FLAGS : Uint8 { MAP_FLAG_NONE = 0, MAP_FLAG_DO_NOT_WAIT = 1, MAP_FLAG_DISCARD = 2, MAP_FLAG_NO_OVERWRITE = 4 };
enum TEXTURE_VIEW_TYPE : Uint8 { TEXTURE_VIEW_UNDEFINED = 0, TEXTURE_VIEW_SHADER_RESOURCE, TEXTURE_VIEW_RENDER_TARGET, TEXTURE_VIEW_DEPTH_STENCIL, TEXTURE_VIEW_READ_ONLY_DEPTH_STENCIL, TEXTURE_VIEW_UNORDERED_ACCESS, TEXTURE_VIEW_NUM_VIEWS };
enum SHADER_TYPE : Uint32 { SHADER_TYPE_UNKNOWN = 0, SHADER_TYPE_VERTEX = 1, SHADER_TYPE_PIXEL = 2, SHADER_TYPE_GEOMETRY = 4, SHADER_TYPE_HULL = 8, SHADER_TYPE_DOMAIN = 16, SHADER_TYPE_COMPUTE = 32, SHADER_TYPE_VS_PS = 3 };
DEFINE_FLAG_ENUM_OPERATORS(SHADER_TYPE)
enum SHADER_SOURCE_LANGUAGE : Uint32 { SHADER_SOURCE_LANGUAGE_DEFAULT = 0, SHADER_SOURCE_LANGUAGE_HLSL, SHADER_SOURCE_LANGUAGE_GLSL };
enum SHADER_COMPILE_FLAGS : Uint32 { SHADER_COMPILE_FLAG_NONE = 0, SHADER_COMPILE_FLAG_ENABLE_UNBOUNDED_ARRAYS = 1, SHADER_COMPILE_FLAG_SKIP_REFLECTION = 2, SHADER_COMPILE_FLAG_ASYNCHRONOUS = 4, SHADER_COMPILE_FLAG_PACK_MATRIX_ROW_MAJOR = 8 };
DEFINE_FLAG_ENUM_OPERATORS(SHADER_COMPILE_FLAGS)
enum PSO_CREATE_FLAGS : Uint32 { PSO_CREATE_FLAG_NONE = 0, PSO_CREATE_FLAG_ASYNCHRONOUS = 8 };
DEFINE_FLAG_ENUM_OPERATORS(PSO_CREATE_FLAGS)
enum PIPELINE_STATE_STATUS : Uint32 { PIPELINE_STATE_STATUS_UNINITIALIZED = 0, PIPELINE_STATE_STATUS_COMPILING, PIPELINE_STATE_STATUS_READY, PIPELINE_STATE_STATUS_FAILED };
enum SHADER_RESOURCE_VARIABLE_TYPE : Uint8 { SHADER_RESOURCE_VARIABLE_TYPE_STATIC = 0, SHADER_RESOURCE_VARIABLE_TYPE_MUTABLE, SHADER_RESOURCE_VARIABLE_TYPE_DYNAMIC, SHADER_RESOURCE_VARIABLE_TYPE_NUM_TYPES };
enum SHADER_VARIABLE_FLAGS : Uint8 { SHADER_VARIABLE_FLAG_NONE = 0, SHADER_VARIABLE_FLAG_NO_DYNAMIC_BUFFERS = 1, SHADER_VARIABLE_FLAG_GENERAL_INPUT_ATTACHMENT_VK = 2, SHADER_VARIABLE_FLAG_UNFILTERABLE_FLOAT_TEXTURE_WEBGPU = 4, SHADER_VARIABLE_FLAG_NON_FILTERING_SAMPLER_WEBGPU = 8 };
DEFINE_FLAG_ENUM_OPERATORS(SHADER_VARIABLE_FLAGS)
enum RESOURCE_STATE_TRANSITION_MODE : Uint8 { RESOURCE_STATE_TRANSITION_MODE_NONE = 0, RESOURCE_STATE_TRANSITION_MODE_TRANSITION, RESOURCE_STATE_TRANSITION_MODE_VERIFY };
enum RESOURCE_STATE : Uint32 { RESOURCE_STATE_UNKNOWN = 0, RESOURCE_STATE_UNDEFINED = 1, RESOURCE_STATE_RENDER_TARGET = 0x10, RESOURCE_STATE_CONSTANT_BUFFER = 0x4, RESOURCE_STATE_SHADER_RESOURCE = 0x80, RESOURCE_STATE_COPY_DEST = 0x400, RESOURCE_STATE_COPY_SOURCE = 0x800 };
enum STATE_TRANSITION_TYPE : Uint8 { STATE_TRANSITION_TYPE_IMMEDIATE = 0, STATE_TRANSITION_TYPE_BEGIN, STATE_TRANSITION_TYPE_END };
enum STATE_TRANSITION_FLAGS : Uint8 { STATE_TRANSITION_FLAG_NONE = 0, STATE_TRANSITION_FLAG_UPDATE_STATE = 1, STATE_TRANSITION_FLAG_DISCARD_CONTENT = 2, STATE_TRANSITION_FLAG_ALIASING = 4 };
DEFINE_FLAG_ENUM_OPERATORS(STATE_TRANSITION_FLAGS)
enum DRAW_FLAGS : Uint8 { DRAW_FLAG_NONE = 0, DRAW_FLAG_VERIFY_STATES = 1, DRAW_FLAG_VERIFY_DRAW_ATTRIBS = 2, DRAW_FLAG_VERIFY_RENDER_TARGETS = 4, DRAW_FLAG_VERIFY_ALL = 7 };
enum VALUE_TYPE : Uint8 { VT_UNDEFINED = 0, VT_INT8, VT_INT16, VT_INT32, VT_UINT8, VT_UINT16, VT_UINT32, VT_FLOAT16, VT_FLOAT32 };
enum CLEAR_DEPTH_STENCIL_FLAGS : Uint32 { CLEAR_DEPTH_FLAG_NONE = 0, CLEAR_DEPTH_FLAG = 1, CLEAR_STENCIL_FLAG = 2 };
DEFINE_FLAG_ENUM_OPERATORS(CLEAR_DEPTH_STENCIL_FLAGS)
enum FILL_MODE : Int8 { FILL_MODE_UNDEFINED = 0, FILL_MODE_WIREFRAME, FILL_MODE_SOLID };
enum CULL_MODE : Int8 { CULL_MODE_UNDEFINED = 0, CULL_MODE_NONE, CULL_MODE_FRONT, CULL_MODE_BACK };
enum PRIMITIVE_TOPOLOGY : Uint8 { PRIMITIVE_TOPOLOGY_UNDEFINED = 0, PRIMITIVE_TOPOLOGY_TRIANGLE_LIST, PRIMITIVE_TOPOLOGY_TRIANGLE_STRIP };
enum FILTER_TYPE : Uint8 { FILTER_TYPE_UNKNOWN = 0, FILTER_TYPE_POINT, FILTER_TYPE_LINEAR, FILTER_TYPE_ANISOTROPIC };
enum TEXTURE_ADDRESS_MODE : Uint8 { TEXTURE_ADDRESS_UNKNOWN = 0, TEXTURE_ADDRESS_WRAP, TEXTURE_ADDRESS_MIRROR, TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_BORDER };
enum COMPARISON_FUNCTION : Uint8 { COMPARISON_FUNC_UNKNOWN = 0, COMPARISON_FUNC_NEVER, COMPARISON_FUNC_LESS, COMPARISON_FUNC_EQUAL, COMPARISON_FUNC_LESS_EQUAL, COMPARISON_FUNC_GREATER, COMPARISON_FUNC_NOT_EQUAL, COMPARISON_FUNC_GREATER_EQUAL, COMPARISON_FUNC_ALWAYS };
inline const char* ComparisonName(COMPARISON_FUNCTION f)
{
    static const char* N[] = {"UNKNOWN", "NEVER", "LESS", "EQUAL", "LESS_EQUAL", "GREATER", "NOT_EQUAL", "GREATER_EQUAL", "ALWAYS"};
    return N[f];
}
enum SET_SHADER_RESOURCE_FLAGS : Uint32 { SET_SHADER_RESOURCE_FLAG_NONE = 0, SET_SHADER_RESOURCE_FLAG_ALLOW_OVERWRITE = 1 };
static constexpr Uint32 REMAINING_MIP_LEVELS   = ~0u;
static constexpr Uint32 REMAINING_ARRAY_SLICES = ~0u;
static constexpr Uint32 DILIGENT_MAX_RENDER_TARGETS = 8;
#define MAX_RENDER_TARGETS 8

// ------------------------------------------------------------------------------------------------ descriptors
struct DeviceObjectAttribs
{
    const Char* Name = nullptr;
};
struct TextureDesc : DeviceObjectAttribs
{
    RESOURCE_DIMENSION Type = RESOURCE_DIM_UNDEFINED;
    Uint32             Width = 0, Height = 0;
    union
    {
        Uint32 ArraySize = 1;
        Uint32 Depth;
    };
    TEXTURE_FORMAT   Format      = TEX_FORMAT_UNKNOWN;
    Uint32           MipLevels   = 1;
    bool             IsCube() const { return Type == RESOURCE_DIM_TEX_CUBE || Type == RESOURCE_DIM_TEX_CUBE_ARRAY; }
    Uint32           SampleCount = 1;
    BIND_FLAGS       BindFlags   = BIND_NONE;
    USAGE            Usage       = USAGE_DEFAULT;
    CPU_ACCESS_FLAGS CPUAccessFlags = CPU_ACCESS_NONE;
    Uint32           MiscFlags   = 0;
    Uint64           ImmediateContextMask = 1;
};
struct TextureSubResData
{
    const void* pData       = nullptr;
    struct IBuffer* pSrcBuffer = nullptr;
    Uint64      SrcOffset   = 0;
    Uint64      Stride      = 0;
    Uint64      DepthStride = 0;
};
struct TextureData
{
    TextureSubResData*     pSubResources   = nullptr;
    Uint32                 NumSubresources = 0;
    struct IDeviceContext* pContext        = nullptr;
};
struct TextureViewDesc : DeviceObjectAttribs
{
    TEXTURE_VIEW_TYPE  ViewType        = TEXTURE_VIEW_UNDEFINED;
    RESOURCE_DIMENSION TextureDim      = RESOURCE_DIM_UNDEFINED;
    TEXTURE_FORMAT     Format          = TEX_FORMAT_UNKNOWN;
    Uint32             MostDetailedMip = 0;
    Uint32             NumMipLevels    = 0;
    Uint32             FirstArraySlice = 0;
    Uint32             NumArraySlices  = 0;
    Uint32             AccessFlags     = 0;
    Uint32             Flags           = 0;
};
struct BufferDesc : DeviceObjectAttribs
{
    Uint64           Size           = 0;
    BIND_FLAGS       BindFlags      = BIND_NONE;
    USAGE            Usage          = USAGE_DEFAULT;
    CPU_ACCESS_FLAGS CPUAccessFlags = CPU_ACCESS_NONE;
    Uint8            Mode           = 0;
    Uint32           ElementByteStride = 0;
    Uint64           ImmediateContextMask = 1;
    constexpr BufferDesc() = default;
    constexpr BufferDesc(const Char* n, Uint64 size, BIND_FLAGS bind, USAGE usage = USAGE_DEFAULT, CPU_ACCESS_FLAGS cpu = CPU_ACCESS_NONE) : Size{size}, BindFlags{bind}, Usage{usage}, CPUAccessFlags{cpu} { Name = n; }
};
struct BufferData
{
    const void* pData    = nullptr;
    Uint64      DataSize = 0;
    struct IDeviceContext* pContext = nullptr;
    constexpr BufferData() = default;
    constexpr BufferData(const void* p, Uint64 n, struct IDeviceContext* c = nullptr) : pData{p}, DataSize{n}, pContext{c} {}
};
struct SamplerDesc : DeviceObjectAttribs
{
    FILTER_TYPE          MinFilter = FILTER_TYPE_LINEAR, MagFilter = FILTER_TYPE_LINEAR, MipFilter = FILTER_TYPE_LINEAR;
    TEXTURE_ADDRESS_MODE AddressU = TEXTURE_ADDRESS_CLAMP, AddressV = TEXTURE_ADDRESS_CLAMP, AddressW = TEXTURE_ADDRESS_CLAMP;
    Uint32               Flags         = 0;
    Bool                 UnnormalizedCoords = false;
    Float32              MipLODBias    = 0;
    Uint32               MaxAnisotropy = 0;
    COMPARISON_FUNCTION  ComparisonFunc = COMPARISON_FUNC_NEVER;
    Float32              BorderColor[4] = {0, 0, 0, 0};
    float                MinLOD = 0, MaxLOD = 3.402823466e+38f;
    constexpr SamplerDesc() = default;
    constexpr SamplerDesc(FILTER_TYPE mn, FILTER_TYPE mg, FILTER_TYPE mp, TEXTURE_ADDRESS_MODE u = TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_MODE v = TEXTURE_ADDRESS_CLAMP, TEXTURE_ADDRESS_MODE w = TEXTURE_ADDRESS_CLAMP) :
        MinFilter{mn}, MagFilter{mg}, MipFilter{mp}, AddressU{u}, AddressV{v}, AddressW{w}
    {}
};
inline std::string SamplerName(const SamplerDesc& s)
{
    static const char* F[] = {"?", "Point", "Linear", "Aniso"};
    static const char* A[] = {"?", "Wrap", "Mirror", "Clamp", "Border"};
    return std::string(F[s.MinFilter]) + A[s.AddressU];
}
struct ShaderMacro
{
    const Char* Name       = nullptr;
    const Char* Definition = nullptr;
};
struct ShaderMacroArray
{
    const ShaderMacro* Elements = nullptr;
    Uint32             Count    = 0;
    constexpr ShaderMacroArray() = default;
    constexpr ShaderMacroArray(const ShaderMacro* e, Uint32 c) : Elements{e}, Count{c} {}
    explicit operator bool() const { return Elements != nullptr && Count > 0; }
};
struct ShaderDesc : DeviceObjectAttribs
{
    SHADER_TYPE ShaderType                 = SHADER_TYPE_UNKNOWN;
    Bool        UseCombinedTextureSamplers = false;
    const Char* CombinedSamplerSuffix      = "_sampler";
    constexpr ShaderDesc() = default;
    constexpr ShaderDesc(const Char* n, SHADER_TYPE t, bool comb = false, const Char* suffix = "_sampler") : ShaderType{t}, UseCombinedTextureSamplers{comb}, CombinedSamplerSuffix{suffix} { Name = n; }
};
struct IShaderSourceInputStreamFactory : IObject
{
};
struct ShaderCreateInfo
{
    const Char*                      FilePath                   = nullptr;
    IShaderSourceInputStreamFactory* pShaderSourceStreamFactory = nullptr;
    const Char*                      Source                     = nullptr;
    const void*                      ByteCode                   = nullptr;
    size_t                           SourceLength               = 0;
    const Char*                      EntryPoint                 = "main";
    ShaderMacroArray                 Macros;
    ShaderDesc                       Desc;
    SHADER_SOURCE_LANGUAGE           SourceLanguage = SHADER_SOURCE_LANGUAGE_DEFAULT;
    Uint32                           ShaderCompiler = 0;
    SHADER_COMPILE_FLAGS             CompileFlags   = SHADER_COMPILE_FLAG_NONE;
    bool                             LoadConstantBufferReflection = false;
};
struct ShaderResourceVariableDesc
{
    const Char*                   Name         = nullptr;
    SHADER_TYPE                   ShaderStages = SHADER_TYPE_UNKNOWN;
    SHADER_RESOURCE_VARIABLE_TYPE Type         = SHADER_RESOURCE_VARIABLE_TYPE_STATIC;
    SHADER_VARIABLE_FLAGS         Flags        = SHADER_VARIABLE_FLAG_NONE;
};
struct ImmutableSamplerDesc
{
    SHADER_TYPE ShaderStages         = SHADER_TYPE_UNKNOWN;
    const Char* SamplerOrTextureName = nullptr;
    SamplerDesc Desc;
};
struct PipelineResourceLayoutDesc
{
    SHADER_RESOURCE_VARIABLE_TYPE     DefaultVariableType        = SHADER_RESOURCE_VARIABLE_TYPE_STATIC;
    SHADER_TYPE                       DefaultVariableMergeStages = SHADER_TYPE_UNKNOWN;
    Uint32                            NumVariables               = 0;
    const ShaderResourceVariableDesc* Variables                  = nullptr;
    Uint32                            NumImmutableSamplers       = 0;
    const ImmutableSamplerDesc*       ImmutableSamplers          = nullptr;
};
struct StencilOpDesc
{
    Uint8 a = 0;
};
struct DepthStencilStateDesc
{
    Bool                DepthEnable      = true;
    Bool                DepthWriteEnable = true;
    COMPARISON_FUNCTION DepthFunc        = COMPARISON_FUNC_LESS;
    Bool                StencilEnable    = false;
};
enum COLOR_MASK : Uint8 { COLOR_MASK_NONE = 0, COLOR_MASK_ALL = 15 };
struct RenderTargetBlendDesc
{
    Bool       BlendEnable           = false;
    COLOR_MASK RenderTargetWriteMask = COLOR_MASK_ALL;
};
struct BlendStateDesc
{
    Bool                  AlphaToCoverageEnable  = false;
    Bool                  IndependentBlendEnable = false;
    RenderTargetBlendDesc RenderTargets[DILIGENT_MAX_RENDER_TARGETS];
};
struct RasterizerStateDesc
{
    FILL_MODE FillMode              = FILL_MODE_SOLID;
    CULL_MODE CullMode              = CULL_MODE_BACK;
    Bool      FrontCounterClockwise = false;
};
struct GraphicsPipelineDesc
{
    BlendStateDesc        BlendDesc;
    Uint32                SampleMask = 0xFFFFFFFF;
    RasterizerStateDesc   RasterizerDesc;
    DepthStencilStateDesc DepthStencilDesc;
    PRIMITIVE_TOPOLOGY    PrimitiveTopology = PRIMITIVE_TOPOLOGY_TRIANGLE_LIST;
    Uint8                 NumViewports      = 1;
    Uint8                 NumRenderTargets  = 0;
    TEXTURE_FORMAT        RTVFormats[DILIGENT_MAX_RENDER_TARGETS] = {};
    TEXTURE_FORMAT        DSVFormat   = TEX_FORMAT_UNKNOWN;
    bool                  ReadOnlyDSV = false;
};
struct PipelineStateDesc : DeviceObjectAttribs
{
    Uint32                     PipelineType = 0;
    PipelineResourceLayoutDesc ResourceLayout;
};
struct IShader;
struct PipelineStateCreateInfo
{
    PipelineStateDesc PSODesc;
    PSO_CREATE_FLAGS  Flags = PSO_CREATE_FLAG_NONE;
};
struct GraphicsPipelineStateCreateInfo : PipelineStateCreateInfo
{
    GraphicsPipelineDesc GraphicsPipeline;
    IShader*             pVS = nullptr;
    IShader*             pPS = nullptr;
};
struct DeviceFeatures
{
    bool TextureSubresourceViews = true;
};
struct RenderDeviceInfo
{
    RENDER_DEVICE_TYPE Type = RENDER_DEVICE_TYPE_VULKAN; // a device with sub-resource views, sub-resource transitions and a base vertex that reaches SV_VertexID
    DeviceFeatures     Features;
    bool               IsD3DDevice() const { return Type == RENDER_DEVICE_TYPE_D3D11 || Type == RENDER_DEVICE_TYPE_D3D12; }
    bool               IsGLDevice() const { return Type == RENDER_DEVICE_TYPE_GL || Type == RENDER_DEVICE_TYPE_GLES; }
    bool               IsVulkanDevice() const { return Type == RENDER_DEVICE_TYPE_VULKAN; }
    bool               IsMetalDevice() const { return Type == RENDER_DEVICE_TYPE_METAL; }
    bool               IsWebGPUDevice() const { return Type == RENDER_DEVICE_TYPE_WEBGPU; }
};
struct SamplerProperties
{
    Bool BorderSamplingModeSupported = true;
    Uint8 MaxAnisotropy = 16;
    Bool LODBiasSupported = true;
};
struct GraphicsAdapterInfo
{
    Uint32            Vendor = 0;
    SamplerProperties Sampler;
};
struct TextureFormatInfo
{
    bool Supported = true;
};
struct TextureFormatInfoExt
{
    BIND_FLAGS BindFlags  = BIND_SHADER_RESOURCE | BIND_RENDER_TARGET;
    bool       Filterable = true;
};
struct DrawAttribs
{
    Uint32     NumVertices           = 0;
    DRAW_FLAGS Flags                 = DRAW_FLAG_NONE;
    Uint32     NumInstances          = 1;
    Uint32     StartVertexLocation   = 0;
    Uint32     FirstInstanceLocation = 0;
    constexpr DrawAttribs() = default;
    constexpr DrawAttribs(Uint32 nv, DRAW_FLAGS f, Uint32 ni = 1, Uint32 sv = 0, Uint32 fi = 0) : NumVertices{nv}, Flags{f}, NumInstances{ni}, StartVertexLocation{sv}, FirstInstanceLocation{fi} {}
};
struct DrawIndexedAttribs
{
    Uint32     NumIndices   = 0;
    VALUE_TYPE IndexType    = VT_UNDEFINED;
    DRAW_FLAGS Flags        = DRAW_FLAG_NONE;
    Uint32     NumInstances = 1;
    constexpr DrawIndexedAttribs() = default;
    Uint32     FirstIndexLocation = 0;
    constexpr DrawIndexedAttribs(Uint32 n, VALUE_TYPE t, DRAW_FLAGS f, Uint32 ni = 1, Uint32 fi = 0) : NumIndices{n}, IndexType{t}, Flags{f}, NumInstances{ni}, FirstIndexLocation{fi} {}
};
struct Box
{
    Uint32 MinX = 0, MaxX = 0, MinY = 0, MaxY = 0, MinZ = 0, MaxZ = 1;
    Box() = default;
    Box(Uint32 x0, Uint32 x1, Uint32 y0, Uint32 y1, Uint32 z0 = 0, Uint32 z1 = 1) : MinX{x0}, MaxX{x1}, MinY{y0}, MaxY{y1}, MinZ{z0}, MaxZ{z1} {}
};
struct ITexture;
struct IBuffer;
struct CopyTextureAttribs
{
    ITexture*                      pSrcTexture = nullptr;
    Uint32                         SrcMipLevel = 0, SrcSlice = 0;
    const Box*                     pSrcBox = nullptr;
    RESOURCE_STATE_TRANSITION_MODE SrcTextureTransitionMode = RESOURCE_STATE_TRANSITION_MODE_NONE;
    ITexture*                      pDstTexture = nullptr;
    Uint32                         DstMipLevel = 0, DstSlice = 0, DstX = 0, DstY = 0, DstZ = 0;
    RESOURCE_STATE_TRANSITION_MODE DstTextureTransitionMode = RESOURCE_STATE_TRANSITION_MODE_NONE;
};
struct IDeviceObject;
struct StateTransitionDesc
{
    IDeviceObject* pResource = nullptr;
    StateTransitionDesc() = default;
    StateTransitionDesc(ITexture* t, RESOURCE_STATE, RESOURCE_STATE, Uint32 = 0, Uint32 = REMAINING_MIP_LEVELS, Uint32 = 0, Uint32 = REMAINING_ARRAY_SLICES, STATE_TRANSITION_TYPE = STATE_TRANSITION_TYPE_IMMEDIATE,
                        STATE_TRANSITION_FLAGS = STATE_TRANSITION_FLAG_NONE);
    StateTransitionDesc(ITexture* t, RESOURCE_STATE, RESOURCE_STATE, STATE_TRANSITION_FLAGS);
    StateTransitionDesc(IBuffer* b, RESOURCE_STATE, RESOURCE_STATE, STATE_TRANSITION_FLAGS = STATE_TRANSITION_FLAG_NONE);
};

// ------------------------------------------------------------------------------------------------ device objects
struct IDeviceObject : IObject
{
    int         id = 0;
    std::string name;
};
struct ISampler : IDeviceObject
{
    SamplerDesc desc;
};
struct ITextureView;
struct ITexture : IDeviceObject
{
    TextureDesc                              desc;
    std::string                              nameStore;
    std::vector<RefCntAutoPtr<ITextureView>> ownedViews; // default views (weak back-pointers: a view does not keep its texture alive)
    const TextureDesc& GetDesc() const { return desc; }
    ITextureView*      GetDefaultView(TEXTURE_VIEW_TYPE type);
    void               CreateView(const TextureViewDesc& vd, ITextureView** ppView);
    ~ITexture() override;
};
struct ITextureView : IDeviceObject
{
    ITexture*       tex = nullptr; // weak for default views (owned by the texture), strong for views made with CreateView (strongTex)
    RefCntAutoPtr<ITexture> strongTex;
    TextureViewDesc desc;
    int             texId = 0; // kept for the log after the texture is gone
    const TextureViewDesc& GetDesc() const { return desc; }
    ITexture*              GetTexture() const { return tex; }
    void                   SetSampler(ISampler*) {}
};
struct IBufferView;
struct IBuffer : IDeviceObject
{
    BufferDesc           desc;
    std::vector<uint8_t> bytes;
    const BufferDesc&    GetDesc() const { return desc; }
};
struct IBufferView : IDeviceObject
{
    IBuffer* buf = nullptr;
    IBuffer* GetBuffer() const { return buf; }
};
struct IShader : IDeviceObject
{
    std::string                                      file, entry, source;
    SHADER_TYPE                                      type = SHADER_TYPE_UNKNOWN;
    std::vector<std::pair<std::string, std::string>> macros;
};
struct IShaderResourceVariable
{
    struct VarStore* store = nullptr;
    std::string      name;
    void             Set(IDeviceObject* obj, SET_SHADER_RESOURCE_FLAGS = SET_SHADER_RESOURCE_FLAG_NONE);
    void             SetArray(IDeviceObject* const* objs, Uint32 first, Uint32 n, SET_SHADER_RESOURCE_FLAGS = SET_SHADER_RESOURCE_FLAG_NONE);
};
struct VarStore // bound objects by variable name (strong references, like a resource binding / a pipeline's static resource cache)
{
    std::map<std::string, RefCntAutoPtr<IDeviceObject>> bound;
    std::map<std::string, IShaderResourceVariable>      vars;
    IShaderResourceVariable* Var(const std::string& n)
    {
        auto it = vars.find(n);
        if (it == vars.end())
        {
            IShaderResourceVariable v;
            v.store = this;
            v.name  = n;
            it      = vars.emplace(n, v).first;
        }
        return &it->second;
    }
};
struct IShaderResourceBinding;
struct IPipelineState : IDeviceObject
{
    RefCntAutoPtr<IShader>                                 vs, ps;
    std::map<std::string, SHADER_RESOURCE_VARIABLE_TYPE>   varTypes;
    SHADER_RESOURCE_VARIABLE_TYPE                          defaultType = SHADER_RESOURCE_VARIABLE_TYPE_STATIC;
    std::map<std::string, std::string>                     immutableSamplers;
    std::vector<TEXTURE_FORMAT>                            rtvFormats;
    DepthStencilStateDesc                                  depth;
    VarStore                                               statics;
    PipelineStateDesc                                      desc;
    PIPELINE_STATE_STATUS GetStatus(bool = false) const { return PIPELINE_STATE_STATUS_READY; }
    const PipelineStateDesc& GetDesc() const { return desc; }
    SHADER_RESOURCE_VARIABLE_TYPE TypeOf(const std::string& n) const
    {
        auto it = varTypes.find(n);
        return it == varTypes.end() ? defaultType : it->second;
    }
    IShaderResourceVariable* GetStaticVariableByName(SHADER_TYPE, const Char* n) { return TypeOf(n) == SHADER_RESOURCE_VARIABLE_TYPE_STATIC ? statics.Var(n) : nullptr; }
    void                     CreateShaderResourceBinding(IShaderResourceBinding** ppSRB, bool InitStaticResources = false);
};
struct IShaderResourceBinding : IDeviceObject
{
    RefCntAutoPtr<IPipelineState> pso;
    VarStore                      vars;
    VarStore                      statics; // the pipeline's static resources as of the creation of the binding (InitStaticResources): a binding carries them along
    IPipelineState*          GetPipelineState() const { return pso; }
    IShaderResourceVariable* GetVariableByName(SHADER_TYPE, const Char* n) { return pso->TypeOf(n) != SHADER_RESOURCE_VARIABLE_TYPE_STATIC ? vars.Var(n) : nullptr; }
};
inline void IPipelineState::CreateShaderResourceBinding(IShaderResourceBinding** ppSRB, bool InitStaticResources)
{
    auto* b = new IShaderResourceBinding();
    b->id   = Recorder::Get().nextId++;
    b->pso  = this;
    if (InitStaticResources) b->statics.bound = statics.bound;
    b->AddRef();
    *ppSRB = b;
}
inline void IShaderResourceVariable::Set(IDeviceObject* obj, SET_SHADER_RESOURCE_FLAGS) { store->bound[name] = obj; }
inline void IShaderResourceVariable::SetArray(IDeviceObject* const* objs, Uint32 first, Uint32 n, SET_SHADER_RESOURCE_FLAGS)
{
    for (Uint32 i = 0; i < n; ++i) store->bound[name + "[" + std::to_string(first + i) + "]"] = objs[i];
}

inline std::string ViewJson(const ITextureView* v)
{
    if (!v) return "null";
    std::ostringstream o;
    o << "{\"tex\":" << v->texId << ",\"mip\":" << v->desc.MostDetailedMip << ",\"mips\":" << v->desc.NumMipLevels << "}";
    return o.str();
}
inline ITextureView* MakeView(ITexture* t, const TextureViewDesc& vdIn, bool strong)
{
    auto* v = new ITextureView();
    v->id   = Recorder::Get().nextId++;
    v->tex  = t;
    v->texId = t->id;
    if (strong) v->strongTex = t;
    v->desc = vdIn;
    if (v->desc.Format == TEX_FORMAT_UNKNOWN) v->desc.Format = t->desc.Format;
    if (v->desc.NumMipLevels == 0 || v->desc.NumMipLevels == REMAINING_MIP_LEVELS)
        v->desc.NumMipLevels = v->desc.ViewType == TEXTURE_VIEW_SHADER_RESOURCE ? t->desc.MipLevels - v->desc.MostDetailedMip : 1;
    return v;
}
inline ITextureView* ITexture::GetDefaultView(TEXTURE_VIEW_TYPE type)
{
    for (auto& v : ownedViews)
        if (v->desc.ViewType == type) return v;
    TextureViewDesc vd;
    vd.ViewType = type;
    ownedViews.emplace_back(MakeView(this, vd, false));
    return ownedViews.back();
}
inline void ITexture::CreateView(const TextureViewDesc& vd, ITextureView** ppView)
{
    ITextureView* v = MakeView(this, vd, true);
    v->AddRef();
    *ppView = v;
}
inline ITexture::~ITexture()
{
    for (auto& v : ownedViews) v->tex = nullptr;
    std::ostringstream o;
    o << "{\"op\":\"destroy_texture\",\"id\":" << id << "}";
    Recorder::Get().Emit(o.str());
}
inline StateTransitionDesc::StateTransitionDesc(ITexture* t, RESOURCE_STATE, RESOURCE_STATE, Uint32, Uint32, Uint32, Uint32, STATE_TRANSITION_TYPE, STATE_TRANSITION_FLAGS) : pResource{t} {}
inline StateTransitionDesc::StateTransitionDesc(ITexture* t, RESOURCE_STATE, RESOURCE_STATE, STATE_TRANSITION_FLAGS) : pResource{t} {}
inline StateTransitionDesc::StateTransitionDesc(IBuffer* b, RESOURCE_STATE, RESOURCE_STATE, STATE_TRANSITION_FLAGS) : pResource{b} {}

// ------------------------------------------------------------------------------------------------ render device
struct IRenderStateCache : IObject
{
};
struct IRenderDevice : IObject
{
    RenderDeviceInfo    info;
    GraphicsAdapterInfo adapter;
    const RenderDeviceInfo&    GetDeviceInfo() const { return info; }
    const GraphicsAdapterInfo& GetAdapterInfo() const { return adapter; }
    TextureFormatInfoExt       GetTextureFormatInfoExt(TEXTURE_FORMAT) const { return TextureFormatInfoExt{}; }
    TextureFormatInfo          GetTextureFormatInfo(TEXTURE_FORMAT) const { return TextureFormatInfo{}; }
    void CreateTexture(const TextureDesc& d, const TextureData* data, ITexture** ppTex)
    {
        auto* t      = new ITexture();
        t->id        = Recorder::Get().nextId++;
        t->nameStore = d.Name ? d.Name : "";
        t->name      = t->nameStore;
        t->desc      = d;
        t->desc.Name = t->nameStore.c_str();
        if (t->desc.MipLevels == 0)
        {
            Uint32 m = 1;
            for (Uint32 s = std::max(d.Width, d.Height); s > 1; s >>= 1) ++m;
            t->desc.MipLevels = m;
        }
        std::ostringstream o;
        o << "{\"op\":\"create_texture\",\"id\":" << t->id << ",\"name\":" << JsonStr(t->name.c_str()) << ",\"w\":" << d.Width << ",\"h\":" << d.Height << ",\"mips\":" << t->desc.MipLevels << ",\"format\":\""
          << FormatName(d.Format) << "\"";
        if (data && data->NumSubresources > 0 && data->pSubResources[0].pData)
        {
            const size_t texel = TexelBytes(d.Format);
            std::string  bytes;
            for (Uint32 y = 0; y < d.Height; ++y) bytes.append(static_cast<const char*>(data->pSubResources[0].pData) + size_t(y) * data->pSubResources[0].Stride, size_t(d.Width) * texel);
            o << ",\"data_b64\":\"" << Base64(bytes.data(), bytes.size()) << "\"";
        }
        o << "}";
        Recorder::Get().Emit(o.str());
        t->AddRef();
        *ppTex = t;
    }
    void CreateBuffer(const BufferDesc& d, const BufferData* data, IBuffer** ppBuf)
    {
        auto* b = new IBuffer();
        b->id   = Recorder::Get().nextId++;
        b->name = d.Name ? d.Name : "";
        b->desc = d;
        b->bytes.assign(size_t(d.Size), 0);
        if (data && data->pData) std::memcpy(b->bytes.data(), data->pData, std::min<size_t>(size_t(d.Size), size_t(data->DataSize)));
        std::ostringstream o;
        o << "{\"op\":\"create_buffer\",\"id\":" << b->id << ",\"name\":" << JsonStr(b->name.c_str()) << ",\"size\":" << d.Size << ",\"bytes_b64\":\"" << Base64(b->bytes.data(), b->bytes.size()) << "\"}";
        Recorder::Get().Emit(o.str());
        b->AddRef();
        *ppBuf = b;
    }
    void CreateShader(const ShaderCreateInfo& ci, IShader** ppShader, void* = nullptr)
    {
        auto* s  = new IShader();
        s->id    = Recorder::Get().nextId++;
        s->name  = ci.Desc.Name ? ci.Desc.Name : "";
        s->file  = ci.FilePath ? ci.FilePath : "";
        s->entry = ci.EntryPoint ? ci.EntryPoint : "";
        s->type  = ci.Desc.ShaderType;
        for (Uint32 i = 0; i < ci.Macros.Count; ++i) s->macros.emplace_back(ci.Macros.Elements[i].Name, ci.Macros.Elements[i].Definition ? ci.Macros.Elements[i].Definition : "");
        s->AddRef();
        *ppShader = s;
    }
    void CreateSampler(const SamplerDesc& d, ISampler** pp)
    {
        auto* s = new ISampler();
        s->id   = Recorder::Get().nextId++;
        s->desc = d;
        s->AddRef();
        *pp = s;
    }
    void CreateGraphicsPipelineState(const GraphicsPipelineStateCreateInfo& ci, IPipelineState** ppPSO)
    {
        auto* p        = new IPipelineState();
        p->id          = Recorder::Get().nextId++;
        p->name        = ci.PSODesc.Name ? ci.PSODesc.Name : "";
        p->desc.Name   = p->name.c_str();
        p->vs          = ci.pVS;
        p->ps          = ci.pPS;
        p->defaultType = ci.PSODesc.ResourceLayout.DefaultVariableType;
        for (Uint32 i = 0; i < ci.PSODesc.ResourceLayout.NumVariables; ++i) p->varTypes[ci.PSODesc.ResourceLayout.Variables[i].Name] = ci.PSODesc.ResourceLayout.Variables[i].Type;
        for (Uint32 i = 0; i < ci.PSODesc.ResourceLayout.NumImmutableSamplers; ++i)
            p->immutableSamplers[ci.PSODesc.ResourceLayout.ImmutableSamplers[i].SamplerOrTextureName] = SamplerName(ci.PSODesc.ResourceLayout.ImmutableSamplers[i].Desc);
        for (Uint32 i = 0; i < ci.GraphicsPipeline.NumRenderTargets; ++i) p->rtvFormats.push_back(ci.GraphicsPipeline.RTVFormats[i]);
        p->depth = ci.GraphicsPipeline.DepthStencilDesc;
        p->AddRef();
        *ppPSO = p;
    }
};

// ------------------------------------------------------------------------------------------------ device context
struct IDeviceContext : IObject
{
    std::vector<RefCntAutoPtr<ITextureView>> rtvs;
    RefCntAutoPtr<ITextureView>              dsv;
    RefCntAutoPtr<IPipelineState>            pso;
    RefCntAutoPtr<IShaderResourceBinding>    srb;
    static std::string Groups()
    {
        std::string g = "[";
        for (size_t i = 0; i < Recorder::Get().groups.size(); ++i) g += (i ? "," : "") + JsonStr(Recorder::Get().groups[i].c_str());
        return g + "]";
    }
    void SetRenderTargets(Uint32 n, ITextureView* const* pp, ITextureView* dsv, RESOURCE_STATE_TRANSITION_MODE)
    {
        rtvs.clear();
        for (Uint32 i = 0; i < n; ++i) rtvs.emplace_back(pp[i]);
        this->dsv = dsv; // (ScreenSpaceReflection keeps its reflection mask in a depth buffer: written by one pass, tested by the others)
    }
    void SetPipelineState(IPipelineState* p) { pso = p; }
    void CommitShaderResources(IShaderResourceBinding* b, RESOURCE_STATE_TRANSITION_MODE) { srb = b; }
    void TransitionResourceStates(Uint32, const StateTransitionDesc*) {}
    void TransitionResourceState(const StateTransitionDesc&) {}
    void SetIndexBuffer(IBuffer*, Uint64, RESOURCE_STATE_TRANSITION_MODE) {}
    void BeginDebugGroup(const Char* n, const float* = nullptr) { Recorder::Get().groups.push_back(n ? n : ""); }
    void EndDebugGroup() { Recorder::Get().groups.pop_back(); }
    void EmitDraw(Uint32 numInstances, Uint32 startVertex)
    {
        std::ostringstream o;
        o << "{\"op\":\"draw\",\"pso\":" << JsonStr(pso ? pso->name.c_str() : "") << ",\"groups\":" << Groups();
        auto macros = [&](IShader* s) {
            o << "{";
            for (size_t i = 0; i < s->macros.size(); ++i) o << (i ? "," : "") << JsonStr(s->macros[i].first.c_str()) << ":" << JsonStr(s->macros[i].second.c_str());
            o << "}";
        };
        if (pso && pso->vs)
        {
            o << ",\"vs\":{\"file\":" << JsonStr(pso->vs->file.c_str()) << ",\"entry\":" << JsonStr(pso->vs->entry.c_str()) << ",\"macros\":";
            macros(pso->vs);
            o << "}";
        }
        if (pso) o << ",\"depth\":{\"enable\":" << (pso->depth.DepthEnable ? "true" : "false") << ",\"write\":" << (pso->depth.DepthWriteEnable ? "true" : "false") << ",\"func\":\"" << ComparisonName(pso->depth.DepthFunc) << "\"}";
        o << ",\"dsv\":" << ViewJson(dsv);
        if (pso && pso->ps)
        {
            o << ",\"ps\":{\"file\":" << JsonStr(pso->ps->file.c_str()) << ",\"entry\":" << JsonStr(pso->ps->entry.c_str()) << ",\"name\":" << JsonStr(pso->ps->name.c_str()) << ",\"macros\":{";
            for (size_t i = 0; i < pso->ps->macros.size(); ++i) o << (i ? "," : "") << JsonStr(pso->ps->macros[i].first.c_str()) << ":" << JsonStr(pso->ps->macros[i].second.c_str());
            o << "}}";
        }
        o << ",\"rtvs\":[";
        for (size_t i = 0; i < rtvs.size(); ++i) o << (i ? "," : "") << ViewJson(rtvs[i]);
        o << "],\"vars\":{";
        bool first = true;
        auto dump  = [&](const VarStore& vs) {
            for (const auto& kv : vs.bound)
            {
                o << (first ? "" : ",") << JsonStr(kv.first.c_str()) << ":";
                first = false;
                IDeviceObject* obj = kv.second;
                if (auto* v = dynamic_cast<ITextureView*>(obj)) o << ViewJson(v);
                else if (auto* b = dynamic_cast<IBuffer*>(obj)) o << "{\"buf\":" << b->id << "}";
                else if (auto* bv = dynamic_cast<IBufferView*>(obj)) o << "{\"buf\":" << (bv->buf ? bv->buf->id : 0) << "}";
                else o << "null";
            }
        };
        // static resources: those the binding was initialised with, else the pipeline's own; a binding of another pipeline with the same resource layout is
        // compatible (TemporalAntiAliasing keeps ONE binding per accumulation buffer across its flag sets)
        if (srb && !srb->statics.bound.empty()) dump(srb->statics);
        else if (pso) dump(pso->statics);
        if (srb) dump(srb->vars);
        o << "},\"samplers\":{";
        if (pso)
        {
            bool f2 = true;
            for (const auto& kv : pso->immutableSamplers)
            {
                o << (f2 ? "" : ",") << JsonStr(kv.first.c_str()) << ":" << JsonStr(kv.second.c_str());
                f2 = false;
            }
        }
        o << "},\"instances\":" << numInstances << ",\"start_vertex\":" << startVertex << "}";
        Recorder::Get().Emit(o.str());
    }
    void Draw(const DrawAttribs& a) { EmitDraw(a.NumInstances, a.StartVertexLocation); }
    void DrawIndexed(const DrawIndexedAttribs& a) { EmitDraw(a.NumInstances, 0); }
    void ClearRenderTarget(ITextureView* v, const void* rgba, RESOURCE_STATE_TRANSITION_MODE)
    {
        const float*       c = static_cast<const float*>(rgba);
        static const float zero[4] = {0, 0, 0, 0};
        if (!c) c = zero;
        std::ostringstream o;
        o.precision(9);
        o << "{\"op\":\"clear\",\"groups\":" << Groups() << ",\"view\":" << ViewJson(v) << ",\"color\":[" << c[0] << "," << c[1] << "," << c[2] << "," << c[3] << "]}";
        Recorder::Get().Emit(o.str());
    }
    void ClearDepthStencil(ITextureView* v, CLEAR_DEPTH_STENCIL_FLAGS, float depth, Uint8, RESOURCE_STATE_TRANSITION_MODE)
    {
        std::ostringstream o;
        o << "{\"op\":\"clear_depth\",\"view\":" << ViewJson(v) << ",\"depth\":" << depth << "}";
        Recorder::Get().Emit(o.str());
    }
    void CopyTexture(const CopyTextureAttribs& a)
    {
        std::ostringstream o;
        o << "{\"op\":\"copy\",\"groups\":" << Groups() << ",\"src\":" << (a.pSrcTexture ? a.pSrcTexture->id : 0) << ",\"src_mip\":" << a.SrcMipLevel << ",\"dst\":" << (a.pDstTexture ? a.pDstTexture->id : 0)
          << ",\"dst_mip\":" << a.DstMipLevel << "}";
        Recorder::Get().Emit(o.str());
    }
    void UpdateTexture(ITexture* t, Uint32 mip, Uint32 slice, const Box& box, const TextureSubResData& data, RESOURCE_STATE_TRANSITION_MODE, RESOURCE_STATE_TRANSITION_MODE)
    {
        const size_t texel = TexelBytes(t->desc.Format);
        std::string  bytes;
        for (Uint32 y = box.MinY; y < box.MaxY; ++y) bytes.append(static_cast<const char*>(data.pData) + size_t(y - box.MinY) * data.Stride, size_t(box.MaxX - box.MinX) * texel);
        std::ostringstream o;
        o << "{\"op\":\"update_texture\",\"tex\":" << t->id << ",\"mip\":" << mip << ",\"slice\":" << slice << ",\"box\":[" << box.MinX << "," << box.MaxX << "," << box.MinY << "," << box.MaxY
          << "],\"data_b64\":\"" << Base64(bytes.data(), bytes.size()) << "\"}";
        Recorder::Get().Emit(o.str());
    }
    void EmitBuffer(IBuffer* b)
    {
        std::ostringstream o;
        o << "{\"op\":\"update_buffer\",\"buf\":" << b->id << ",\"name\":" << JsonStr(b->name.c_str()) << ",\"bytes_b64\":\"" << Base64(b->bytes.data(), b->bytes.size()) << "\"}";
        Recorder::Get().Emit(o.str());
    }
    void UpdateBuffer(IBuffer* b, Uint64 offset, Uint64 size, const void* data, RESOURCE_STATE_TRANSITION_MODE)
    {
        if (offset + size > b->bytes.size()) { Recorder::Get().Error("UpdateBuffer: out of range"); return; }
        std::memcpy(b->bytes.data() + offset, data, size_t(size));
        EmitBuffer(b);
    }
    void MapBuffer(IBuffer* b, MAP_TYPE, MAP_FLAGS, void*& p) { p = b->bytes.data(); }
    void UnmapBuffer(IBuffer* b, MAP_TYPE) { EmitBuffer(b); }
};

} // namespace Diligent
