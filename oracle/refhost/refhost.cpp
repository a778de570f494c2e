// refhost.cpp -- TEST INFRASTRUCTURE ONLY (oracle/_ref/libmifx_refhost.so; never part of the product).
//
// Drives the reference's own host classes -- PostFXContext, ScreenSpaceAmbientOcclusion, ScreenSpaceReflection, TemporalAntiAliasing, Bloom, compiled from the sources where
// they lie under /root/reference/PostProcess against the recording DiligentCore stand-in (dg/flat/dg_mock.hpp) -- through the call protocol of their only in-repo caller
// (Hydrogent/src/Tasks/HnPostProcessTask.cpp:671-682 Prepare, :788-918 Execute) and returns what they asked the device to do as a list of JSON commands.
// oracle/refhost.py turns the list into a frame by running the reference's shaders (oracle/_ref) pass by pass; tests/test_host_sequence_vs_ref.py compares the result with
// oracle/cpu_chain.py, the hand restatement of this sequencing that the product's csrc/api_*.cpp follow.
#include <cstring>
#include <memory>
#include <string>

#include "PostProcess/Common/interface/PostFXContext.hpp"
#include "ScreenSpaceAmbientOcclusion.hpp"
#include "ScreenSpaceReflection.hpp"
#include "TemporalAntiAliasing.hpp"
#include "Bloom.hpp"
#include "DepthOfField.hpp"
#include "Components/interface/EnvMapRenderer.hpp"
#include "Utilities/interface/DiligentFXShaderSourceStreamFactory.hpp"

namespace Diligent
{
namespace HLSL
{
#include "Shaders/Common/public/BasicStructures.fxh"
#include "Shaders/PostProcess/ScreenSpaceAmbientOcclusion/public/ScreenSpaceAmbientOcclusionStructures.fxh"
#include "Shaders/PostProcess/ScreenSpaceReflection/public/ScreenSpaceReflectionStructures.fxh"
#include "Shaders/PostProcess/TemporalAntiAliasing/public/TemporalAntiAliasingStructures.fxh"
#include "Shaders/PostProcess/Bloom/public/BloomStructures.fxh"
#include "Shaders/PostProcess/DepthOfField/public/DepthOfFieldStructures.fxh"
#include "Shaders/PostProcess/ToneMapping/public/ToneMappingStructures.fxh"
} // namespace HLSL

// Utilities/src/DiligentFXShaderSourceStreamFactory.cpp is not compiled (it loads shader files through DiligentCore); the post-process classes only pass the instance on
DiligentFXShaderSourceStreamFactory::DiligentFXShaderSourceStreamFactory() {}
IShaderSourceInputStreamFactory& DiligentFXShaderSourceStreamFactory::GetInstance()
{
    static IShaderSourceInputStreamFactory f;
    return f;
}
} // namespace Diligent

using namespace Diligent;

namespace
{
struct Host
{
    RefCntAutoPtr<IRenderDevice>  device;
    RefCntAutoPtr<IDeviceContext> context;
    std::unique_ptr<PostFXContext>               postfx;
    std::unique_ptr<ScreenSpaceAmbientOcclusion> ssao;
    std::unique_ptr<ScreenSpaceReflection>       ssr;
    std::unique_ptr<TemporalAntiAliasing>        taa;
    std::unique_ptr<Bloom>                       bloom;
    std::unique_ptr<DepthOfField>                dof;
    // the caller's frame inputs: textures recreated when the frame size changes (the application owns them in the reference)
    Uint32 w = 0, h = 0;
    RefCntAutoPtr<ITexture> depth, prevDepth, motion, normal, material, color, composite;
    std::string out;
};

RefCntAutoPtr<ITexture> make_input(Host* host, const char* name, Uint32 w, Uint32 h, TEXTURE_FORMAT fmt)
{
    TextureDesc d;
    d.Name = name; d.Type = RESOURCE_DIM_TEX_2D; d.Width = w; d.Height = h; d.Format = fmt; d.MipLevels = 1; d.BindFlags = BIND_SHADER_RESOURCE | BIND_RENDER_TARGET;
    RefCntAutoPtr<ITexture> t;
    host->device->CreateTexture(d, nullptr, &t);
    return t;
}
std::string flush(Host* host)
{
    std::string s = "[";
    auto& lines = Recorder::Get().lines;
    for (size_t i = 0; i < lines.size(); ++i) s += (i ? ",\n" : "") + lines[i];
    lines.clear();
    return s + "]";
}
} // namespace

extern "C" {

// which effects the host holds: bit 0 SSAO, 1 SSR, 2 TAA, 3 Bloom, 4 depth of field (the PostFX context always)
void* refhost_create(unsigned effects)
{
    Recorder::Get() = Recorder{};
    Host* host = new Host();
    host->device  = RefCntAutoPtr<IRenderDevice>{new IRenderDevice()};
    host->context = RefCntAutoPtr<IDeviceContext>{new IDeviceContext()};
    host->postfx  = std::make_unique<PostFXContext>(host->device, PostFXContext::CreateInfo{});
    if (effects & 1u) host->ssao = std::make_unique<ScreenSpaceAmbientOcclusion>(host->device, ScreenSpaceAmbientOcclusion::CreateInfo{});
    if (effects & 2u) host->ssr = std::make_unique<ScreenSpaceReflection>(host->device, ScreenSpaceReflection::CreateInfo{});
    if (effects & 4u) host->taa = std::make_unique<TemporalAntiAliasing>(host->device, TemporalAntiAliasing::CreateInfo{});
    if (effects & 8u) host->bloom = std::make_unique<Bloom>(host->device, Bloom::CreateInfo{});
    if (effects & 16u) host->dof = std::make_unique<DepthOfField>(host->device, DepthOfField::CreateInfo{});
    return host;
}
void refhost_destroy(void* p) { delete static_cast<Host*>(p); }

struct refhost_frame
{
    unsigned    index, width, height;
    unsigned    postfx_flags, ssao_flags, ssr_flags, taa_flags, bloom_flags;
    float       timer_elapsed;   // what the effects' frame timers read: AlphaInterpolation = clamp(elapsed, 0, 1) (ScreenSpaceAmbientOcclusion.cpp:790-795)
    const void* curr_camera;     // HLSL::CameraAttribs
    const void* prev_camera;
    const void* ssao_attribs;    // HLSL::ScreenSpaceAmbientOcclusionAttribs, ... (NULL: the effect is not executed this frame)
    const void* ssr_attribs;
    const void* taa_attribs;
    const void* bloom_attribs;
    unsigned    dof_flags;
    const void* dof_attribs;     // HLSL::DepthOfFieldAttribs
};

// One frame in the order of HnPostProcessTask: PrepareResources of PostFX, SSAO, SSR, TAA, Bloom (:671-682); PostFXContext::Execute (:788-809), SSR (:811-822), SSAO (:824-832),
// [the application's composite into `composite`: recorded as {"op":"app_composite"}], TAA on it (:871-897), depth of field on the TAA output (:899-909), Bloom on the
// result (:911-918).
// Returns the JSON command list of the frame (valid until the next call on this host).
const char* refhost_frame_execute(void* p, const refhost_frame* f)
{
    Host* host = static_cast<Host*>(p);
    Recorder::Get().timerElapsed = f->timer_elapsed;
    if (host->w != f->width || host->h != f->height)
    {
        host->w = f->width; host->h = f->height;
        host->depth     = make_input(host, "input::depth", f->width, f->height, TEX_FORMAT_R32_FLOAT);
        host->prevDepth = make_input(host, "input::prev_depth", f->width, f->height, TEX_FORMAT_R32_FLOAT);
        host->motion    = make_input(host, "input::motion", f->width, f->height, TEX_FORMAT_RG16_FLOAT);
        host->normal    = make_input(host, "input::normal", f->width, f->height, TEX_FORMAT_RGBA16_FLOAT);
        host->material  = make_input(host, "input::material", f->width, f->height, TEX_FORMAT_RGBA16_FLOAT);
        host->color     = make_input(host, "input::color", f->width, f->height, TEX_FORMAT_RGBA16_FLOAT);
        host->composite = make_input(host, "input::composite", f->width, f->height, TEX_FORMAT_RGBA16_FLOAT);
    }
    IRenderDevice*  dev = host->device;
    IDeviceContext* ctx = host->context;
    {
        std::string s = "{\"op\":\"frame\",\"index\":" + std::to_string(f->index) + ",\"inputs\":{";
        const char*  names[] = {"depth", "prev_depth", "motion", "normal", "material", "color", "composite"};
        ITexture*    tex[]   = {host->depth, host->prevDepth, host->motion, host->normal, host->material, host->color, host->composite};
        for (int i = 0; i < 7; ++i) s += std::string(i ? "," : "") + "\"" + names[i] + "\":" + std::to_string(tex[i]->id);
        Recorder::Get().Emit(s + "}}");
    }
    // Prepare
    PostFXContext::FrameDesc fd;
    fd.Index = f->index; fd.Width = f->width; fd.Height = f->height; fd.OutputWidth = f->width; fd.OutputHeight = f->height;
    host->postfx->PrepareResources(dev, fd, static_cast<PostFXContext::FEATURE_FLAGS>(f->postfx_flags));
    if (host->ssao) host->ssao->PrepareResources(dev, ctx, host->postfx.get(), static_cast<ScreenSpaceAmbientOcclusion::FEATURE_FLAGS>(f->ssao_flags));
    if (host->ssr) host->ssr->PrepareResources(dev, ctx, host->postfx.get(), static_cast<ScreenSpaceReflection::FEATURE_FLAGS>(f->ssr_flags));
    if (host->taa) host->taa->PrepareResources(dev, ctx, host->postfx.get(), static_cast<TemporalAntiAliasing::FEATURE_FLAGS>(f->taa_flags));
    if (host->bloom) host->bloom->PrepareResources(dev, ctx, host->postfx.get(), static_cast<Bloom::FEATURE_FLAGS>(f->bloom_flags));
    if (host->dof) host->dof->PrepareResources(dev, ctx, host->postfx.get(), static_cast<DepthOfField::FEATURE_FLAGS>(f->dof_flags)); // (:680-683, after Bloom's)
    // Execute
    {
        PostFXContext::RenderAttributes a;
        a.pDevice = dev; a.pDeviceContext = ctx;
        a.pCurrDepthBufferSRV = host->depth->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pPrevDepthBufferSRV = host->prevDepth->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pMotionVectorsSRV   = host->motion->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pCurrCamera = static_cast<const HLSL::CameraAttribs*>(f->curr_camera);
        a.pPrevCamera = static_cast<const HLSL::CameraAttribs*>(f->prev_camera);
        host->postfx->Execute(a);
    }
    if (host->ssr && f->ssr_attribs)
    {
        ScreenSpaceReflection::RenderAttributes a;
        a.pDevice = dev; a.pDeviceContext = ctx; a.pPostFXContext = host->postfx.get();
        a.pColorBufferSRV    = host->color->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pDepthBufferSRV    = host->depth->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pNormalBufferSRV   = host->normal->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pMaterialBufferSRV = host->material->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pMotionVectorsSRV  = host->motion->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pSSRAttribs        = static_cast<const HLSL::ScreenSpaceReflectionAttribs*>(f->ssr_attribs);
        host->ssr->Execute(a);
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"ssr\",\"view\":" + ViewJson(host->ssr->GetSSRRadianceSRV()) + "}");
    }
    if (host->ssao && f->ssao_attribs)
    {
        ScreenSpaceAmbientOcclusion::RenderAttributes a;
        a.pDevice = dev; a.pDeviceContext = ctx; a.pPostFXContext = host->postfx.get();
        a.pDepthBufferSRV  = host->depth->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pNormalBufferSRV = host->normal->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pSSAOAttribs     = static_cast<const HLSL::ScreenSpaceAmbientOcclusionAttribs*>(f->ssao_attribs);
        host->ssao->Execute(a);
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"ssao\",\"view\":" + ViewJson(host->ssao->GetAmbientOcclusionSRV()) + "}");
    }
    Recorder::Get().Emit("{\"op\":\"app_composite\",\"dst\":" + std::to_string(host->composite->id) + "}");
    ITextureView* frameSRV = host->composite->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
    if (host->taa && f->taa_attribs)
    {
        TemporalAntiAliasing::RenderAttributes a;
        a.pDevice = dev; a.pDeviceContext = ctx; a.pPostFXContext = host->postfx.get();
        a.pColorBufferSRV = frameSRV;
        a.pTAAAttribs     = static_cast<const HLSL::TemporalAntiAliasingAttribs*>(f->taa_attribs);
        host->taa->Execute(a);
        frameSRV = host->taa->GetAccumulatedFrameSRV();
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"taa\",\"view\":" + ViewJson(frameSRV) + "}");
    }
    if (host->dof && f->dof_attribs)
    {
        DepthOfField::RenderAttributes a;
        a.pDevice = dev; a.pDeviceContext = ctx; a.pPostFXContext = host->postfx.get();
        a.pColorBufferSRV = frameSRV;
        a.pDepthBufferSRV = host->depth->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
        a.pDOFAttribs     = static_cast<const HLSL::DepthOfFieldAttribs*>(f->dof_attribs);
        host->dof->Execute(a);
        frameSRV = host->dof->GetDepthOfFieldTextureSRV();
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"dof\",\"view\":" + ViewJson(frameSRV) + "}");
    }
    if (host->bloom && f->bloom_attribs)
    {
        Bloom::RenderAttributes a;
        a.pDevice = dev; a.pDeviceContext = ctx; a.pPostFXContext = host->postfx.get();
        a.pColorBufferSRV = frameSRV;
        a.pBloomAttribs   = static_cast<const HLSL::BloomAttribs*>(f->bloom_attribs);
        host->bloom->Execute(a);
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"bloom\",\"view\":" + ViewJson(host->bloom->GetBloomTextureSRV()) + "}");
    }
    {
        // the PostFX context's own outputs (what the effects of the NEXT frame and the application read)
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"reprojected_depth\",\"view\":" + ViewJson(host->postfx->GetReprojectedDepth()) + "}");
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"closest_motion\",\"view\":" + ViewJson(host->postfx->GetClosestMotionVectors()) + "}");
        Recorder::Get().Emit("{\"op\":\"output\",\"effect\":\"previous_depth\",\"view\":" + ViewJson(host->postfx->GetPreviousDepth()) + "}");
    }
    host->out = flush(host);
    return host->out.c_str();
}

// Components/src/EnvMapRenderer.cpp (the environment-map background pass), executed: Prepare + Render of one frame with a cube or a sphere map, the render targets
// bound the way its only in-repo caller binds them (colour, motion vectors; depth read-only).  Returns the command list: the constant buffer the class fills, the
// pipeline's macros and depth state, the draw.
struct refhost_envmap
{
    unsigned    width, height, options; // EnvMapRenderer::OPTION_FLAGS
    unsigned    cube, env_size, env_mips;
    float       average_log_lum, mip_level, alpha, scale[3];
    const void* tone_mapping;           // HLSL::ToneMappingAttribs
    const void* cameras;                // two HLSL::CameraAttribs: g_Camera, g_PrevCamera (cbCameraAttribs of EnvMap.psh)
};
const char* refhost_envmap_render(const refhost_envmap* e)
{
    static std::string out;
    Recorder::Get() = Recorder{};
    RefCntAutoPtr<IRenderDevice>  device{new IRenderDevice()};
    RefCntAutoPtr<IDeviceContext> context{new IDeviceContext()};
    RefCntAutoPtr<IBuffer>        cameraCB;
    CreateUniformBuffer(device, 2 * sizeof(HLSL::CameraAttribs), "Camera attribs CB", &cameraCB, USAGE_DEFAULT, BIND_UNIFORM_BUFFER, CPU_ACCESS_NONE, const_cast<void*>(e->cameras));
    auto tex = [&](const char* name, RESOURCE_DIMENSION dim, Uint32 w, Uint32 h, Uint32 slices, Uint32 mips, TEXTURE_FORMAT fmt, BIND_FLAGS bind) {
        TextureDesc d;
        d.Name = name; d.Type = dim; d.Width = w; d.Height = h; d.ArraySize = slices; d.MipLevels = mips; d.Format = fmt; d.BindFlags = bind;
        RefCntAutoPtr<ITexture> t;
        device->CreateTexture(d, nullptr, &t);
        return t;
    };
    RefCntAutoPtr<ITexture> env    = e->cube ? tex("input::env_cube", RESOURCE_DIM_TEX_CUBE, e->env_size, e->env_size, 6, e->env_mips, TEX_FORMAT_RGBA32_FLOAT, BIND_SHADER_RESOURCE)
                                             : tex("input::env_sphere", RESOURCE_DIM_TEX_2D, 2 * e->env_size, e->env_size, 1, e->env_mips, TEX_FORMAT_RGBA32_FLOAT, BIND_SHADER_RESOURCE);
    RefCntAutoPtr<ITexture> color  = tex("target::color", RESOURCE_DIM_TEX_2D, e->width, e->height, 1, 1, TEX_FORMAT_RGBA16_FLOAT, BIND_SHADER_RESOURCE | BIND_RENDER_TARGET);
    RefCntAutoPtr<ITexture> motion = tex("target::motion", RESOURCE_DIM_TEX_2D, e->width, e->height, 1, 1, TEX_FORMAT_RG16_FLOAT, BIND_SHADER_RESOURCE | BIND_RENDER_TARGET);
    RefCntAutoPtr<ITexture> depth  = tex("target::depth", RESOURCE_DIM_TEX_2D, e->width, e->height, 1, 1, TEX_FORMAT_D32_FLOAT, BIND_SHADER_RESOURCE | BIND_DEPTH_STENCIL);
    EnvMapRenderer::CreateInfo ci;
    ci.pDevice = device; ci.pCameraAttribsCB = cameraCB; ci.NumRenderTargets = 2; ci.RTVFormats[0] = TEX_FORMAT_RGBA16_FLOAT; ci.RTVFormats[1] = TEX_FORMAT_RG16_FLOAT;
    ci.DSVFormat = TEX_FORMAT_D32_FLOAT; ci.RenderTargetMask = 0x3u;
    EnvMapRenderer renderer{ci};
    EnvMapRenderer::RenderAttribs ra;
    ra.pEnvMap = env->GetDefaultView(TEXTURE_VIEW_SHADER_RESOURCE);
    ra.AverageLogLum = e->average_log_lum; ra.MipLevel = e->mip_level; ra.Alpha = e->alpha;
    ra.Options = static_cast<EnvMapRenderer::OPTION_FLAGS>(e->options);
    ra.Scale = float3{e->scale[0], e->scale[1], e->scale[2]};
    renderer.Prepare(context, ra, *static_cast<const HLSL::ToneMappingAttribs*>(e->tone_mapping));
    ITextureView* rtvs[] = {color->GetDefaultView(TEXTURE_VIEW_RENDER_TARGET), motion->GetDefaultView(TEXTURE_VIEW_RENDER_TARGET)};
    context->SetRenderTargets(2, rtvs, depth->GetDefaultView(TEXTURE_VIEW_DEPTH_STENCIL), RESOURCE_STATE_TRANSITION_MODE_TRANSITION);
    renderer.Render(context);
    std::string s = "[";
    auto& lines = Recorder::Get().lines;
    for (size_t i = 0; i < lines.size(); ++i) s += (i ? ",\n" : "") + lines[i];
    lines.clear();
    out = s + "]";
    return out.c_str();
}

// TemporalAntiAliasing::GetJitterOffset as the object computes it for the frame it was last prepared for
void refhost_taa_jitter(void* p, float out[2])
{
    Host* host = static_cast<Host*>(p);
    const float2 j = host->taa ? host->taa->GetJitterOffset() : float2{0.0f, 0.0f};
    out[0] = j.x; out[1] = j.y;
}
unsigned refhost_sizeof(const char* what)
{
    const std::string w = what;
    if (w == "CameraAttribs") return sizeof(HLSL::CameraAttribs);
    if (w == "ScreenSpaceAmbientOcclusionAttribs") return sizeof(HLSL::ScreenSpaceAmbientOcclusionAttribs);
    if (w == "ScreenSpaceReflectionAttribs") return sizeof(HLSL::ScreenSpaceReflectionAttribs);
    if (w == "TemporalAntiAliasingAttribs") return sizeof(HLSL::TemporalAntiAliasingAttribs);
    if (w == "BloomAttribs") return sizeof(HLSL::BloomAttribs);
    if (w == "DepthOfFieldAttribs") return sizeof(HLSL::DepthOfFieldAttribs);
    if (w == "ToneMappingAttribs") return sizeof(HLSL::ToneMappingAttribs);
    return 0;
}
} // extern "C"
