// diligent_api_stand_in.hpp -- the few DiligentCore / DiligentFX names the adapters touch, declared just far enough for
// mifx_effect_adapters.cpp to compile OUTSIDE the reference tree (tests/test_adapter_example.py builds it with g++ on the CPU).
// Inside the reference tree this file is not used: the adapters include the real headers instead
// (PostFXContext.hpp, ScreenSpaceAmbientOcclusion.hpp, ..., and the Shaders/**/public/*Structures.fxh attribute blocks).
// Nothing here has behaviour; sizes of the attribute blocks are the ones SURVEY.md Appendix B measured on the reference headers.
#pragma once
#include <cstdint>
#include <cstdio>
#include <sstream>

namespace Diligent
{
using Uint32 = uint32_t;
struct IRenderDevice;
struct IDeviceContext;
struct IRenderStateCache;
struct ITextureView;
struct IBuffer;
struct float2
{
    float x, y;
};
struct float4x4
{
    float m[16]; // row-major, row-vector convention
};

namespace HLSL
{
// opaque here: the adapters only memcpy these blocks into their mifx twins (same bytes, see the static_asserts in the .cpp)
struct CameraAttribs { unsigned char bytes[576]; };
struct ScreenSpaceAmbientOcclusionAttribs { unsigned char bytes[48]; };
struct ScreenSpaceReflectionAttribs { unsigned char bytes[48]; };
struct TemporalAntiAliasingAttribs { unsigned char bytes[16]; };
struct BloomAttribs { unsigned char bytes[32]; };
struct DepthOfFieldAttribs { unsigned char bytes[32]; };
struct ToneMappingAttribs { unsigned char bytes[48]; };
struct PBRMaterialBasicAttribs { unsigned char bytes[96]; };      // Shaders/PBR/public/PBR_Structures.fxh:154-180
struct PBRRendererShaderParameters { unsigned char bytes[144]; }; // :126-149
struct PBRLightAttribs { unsigned char bytes[64]; };              // :309-330
struct PBRShadowMapInfo { unsigned char bytes[96]; };             // :336-347
} // namespace HLSL

namespace NoiseBuffers // PostProcess/Common/src/SamplerBlueNoiseErrorDistribution_128x128_OptimizedFor_2d2d2d2d_1spp.cpp
{
extern const unsigned char Sobol_256d[256];
extern const unsigned char ScramblingTile[128 * 128 * 8];
} // namespace NoiseBuffers

// seconds since the previous Restart(), as Diligent::Timer
struct Timer
{
    float GetElapsedTimef() const { return 1.0f; }
    void  Restart() {}
};
} // namespace Diligent

namespace Diligent
{
template <class... Args> void StandInLog(const char* kind, const Args&... args)
{
    std::ostringstream s;
    (s << ... << args);
    std::fprintf(stderr, "[DiligentFX/mifx] %s: %s\n", kind, s.str().c_str());
}
} // namespace Diligent
#define LOG_ERROR_MESSAGE(...) ::Diligent::StandInLog("error", __VA_ARGS__)
#define DEV_CHECK_ERR(cond, ...) \
    do { if (!(cond)) ::Diligent::StandInLog("check failed", __VA_ARGS__); } while (false)
