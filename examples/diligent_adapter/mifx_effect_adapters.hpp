// mifx_effect_adapters.hpp -- DiligentFX's post-processing classes implemented on the mifx C ABI (include/mifx.h).
//
// What a maintainer of the reference adds to run the effects on an MI355X through HIP instead of through pixel shaders: the public
// interface of every class is the reference's (same method names, argument meaning, call protocol and error behaviour -- methods return
// void, misuse is logged, never thrown); the private part shrinks to one opaque mifx handle.  Reference declarations:
//   PostFXContext                  PostProcess/Common/interface/PostFXContext.hpp:48-263
//   ScreenSpaceAmbientOcclusion    PostProcess/ScreenSpaceAmbientOcclusion/interface/ScreenSpaceAmbientOcclusion.hpp:57-262
//   ScreenSpaceReflection          PostProcess/ScreenSpaceReflection/interface/ScreenSpaceReflection.hpp:62-260
//   TemporalAntiAliasing           PostProcess/TemporalAntiAliasing/interface/TemporalAntiAliasing.hpp:60-215
//   Bloom                          PostProcess/Bloom/interface/Bloom.hpp:58-170
//   DepthOfField                   PostProcess/DepthOfField/interface/DepthOfField.hpp:57-240
// UpdateUI (ImGui) and the PSO / shader caches of the originals have no counterpart here.
// Built on the CPU by tests/test_adapter_example.py (g++, against libmifx.so); see INTEGRATION.md section 1.
#pragma once
#include "mifx_interop.hpp"

namespace Diligent
{
class PostFXContext
{
public:
    enum FEATURE_FLAGS : Uint32
    {
        FEATURE_FLAG_NONE                 = 0u,
        FEATURE_FLAG_REVERSED_DEPTH       = 1u << 0u,
        FEATURE_FLAG_HALF_PRECISION_DEPTH = 1u << 1u,
        FEATURE_FLAG_TEMPORAL_UPSCALING   = 1u << 2u
    };
    enum BLUE_NOISE_DIMENSION : Uint32
    {
        BLUE_NOISE_DIMENSION_XY = 0,
        BLUE_NOISE_DIMENSION_ZW
    };
    struct CreateInfo
    {
        bool EnableAsyncCreation = false; // nothing to create asynchronously: the kernels are in libmifx.so
        bool PackMatrixRowMajor  = false; // the camera block is consumed on the host, row-major as CameraAttribs stores it
    };
    struct FrameDesc
    {
        Uint32 Index = 0, Width = 0, Height = 0, OutputWidth = 0, OutputHeight = 0;
    };
    struct SupportedDeviceFeatures
    {
        bool TransitionSubresources  = false;
        bool TextureSubresourceViews = false;
        bool CopyDepthToColor        = false;
        bool ShaderBaseVertexOffset  = false;
    };
    struct TextureOperationAttribs
    {
        IRenderDevice*     pDevice        = nullptr;
        IRenderStateCache* pStateCache    = nullptr;
        IDeviceContext*    pDeviceContext = nullptr;
    };
    struct RenderAttributes
    {
        IRenderDevice*             pDevice             = nullptr;
        IRenderStateCache*         pStateCache         = nullptr;
        IDeviceContext*            pDeviceContext      = nullptr;
        ITextureView*              pCurrDepthBufferSRV = nullptr;
        ITextureView*              pPrevDepthBufferSRV = nullptr;
        ITextureView*              pMotionVectorsSRV   = nullptr;
        const HLSL::CameraAttribs* pCurrCamera         = nullptr;
        const HLSL::CameraAttribs* pPrevCamera         = nullptr;
        IBuffer*                   pCameraAttribsCB    = nullptr; // not supported: the camera blocks must be passed by pointer
    };

    PostFXContext(IRenderDevice* pDevice, const CreateInfo& CI, int HipDevice = 0);
    ~PostFXContext();

    void PrepareResources(IRenderDevice* pDevice, const FrameDesc& Desc, FEATURE_FLAGS FeatureFlags);
    void Execute(const RenderAttributes& RenderAttribs);

    ITextureView* Get2DBlueNoiseSRV(BLUE_NOISE_DIMENSION Dimension) const;
    ITextureView* GetReprojectedDepth() const;
    ITextureView* GetPreviousDepth() const;
    ITextureView* GetClosestMotionVectors() const;
    const FrameDesc& GetFrameDesc() const { return m_FrameDesc; }
    FEATURE_FLAGS    GetFeatureFlags() const { return m_FeatureFlags; }
    float            GetInterpolationSpeed() const { return 1.0f; }
    SupportedDeviceFeatures GetSupportedFeatures() const;
    // the texture helpers the effects and the application share (PostFXContext.hpp:168-172); the clear takes the view of its target here (a texture's default
    // render-target view in the reference), `Channels` floats of ClearColor are used
    void ClearRenderTarget(const TextureOperationAttribs& Attribs, ITextureView* pRTV, uint32_t Format, float ClearColor[]);
    void CopyTextureDepth(const TextureOperationAttribs& Attribs, ITextureView* pSRV, ITextureView* pRTV);
    void CopyTextureColor(const TextureOperationAttribs& Attribs, ITextureView* pSRV, ITextureView* pRTV);

    mifx_postfx* GetMifxContext() const { return m_Mifx; } // the one addition to the public interface

private:
    mifx_postfx*  m_Mifx = nullptr;
    FrameDesc     m_FrameDesc;
    FEATURE_FLAGS m_FeatureFlags = FEATURE_FLAG_NONE;
};

// The five effects share one shape: create lazily in PrepareResources, copy the attribute block and fill in what the reference's
// UpdateConstantBuffer computes on the host (AlphaInterpolation from the frame timer), execute, hand out the effect-owned output.
class ScreenSpaceAmbientOcclusion
{
public:
    enum FEATURE_FLAGS : Uint32
    {
        FEATURE_FLAG_NONE                 = 0u,
        FEATURE_FLAG_HALF_PRECISION_DEPTH = 1u << 0u,
        FEATURE_FLAG_HALF_RESOLUTION      = 1u << 1u
    };
    struct CreateInfo
    {
        bool EnableAsyncCreation = false;
    };
    struct RenderAttributes
    {
        IRenderDevice*     pDevice          = nullptr;
        IRenderStateCache* pStateCache      = nullptr;
        IDeviceContext*    pDeviceContext   = nullptr;
        PostFXContext*     pPostFXContext   = nullptr;
        ITextureView*      pDepthBufferSRV  = nullptr;
        ITextureView*      pNormalBufferSRV = nullptr;
        const HLSL::ScreenSpaceAmbientOcclusionAttribs* pSSAOAttribs = nullptr;
    };
    ScreenSpaceAmbientOcclusion(IRenderDevice* pDevice, const CreateInfo& CI);
    ~ScreenSpaceAmbientOcclusion();
    void          PrepareResources(IRenderDevice* pDevice, IDeviceContext* pDeviceContext, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags);
    void          Execute(const RenderAttributes& RenderAttribs);
    ITextureView* GetAmbientOcclusionSRV() const;

private:
    mifx_ssao* m_Impl = nullptr;
    Timer      m_FrameTimer;
};

class ScreenSpaceReflection
{
public:
    enum FEATURE_FLAGS : Uint32
    {
        FEATURE_FLAG_NONE            = 0u,
        FEATURE_FLAG_PREVIOUS_FRAME  = 1u << 0u,
        FEATURE_FLAG_HALF_RESOLUTION = 1u << 1u
    };
    struct CreateInfo
    {
        bool EnableAsyncCreation = false;
    };
    struct RenderAttributes
    {
        IRenderDevice*     pDevice            = nullptr;
        IRenderStateCache* pStateCache        = nullptr;
        IDeviceContext*    pDeviceContext     = nullptr;
        PostFXContext*     pPostFXContext     = nullptr;
        ITextureView*      pColorBufferSRV    = nullptr;
        ITextureView*      pDepthBufferSRV    = nullptr;
        ITextureView*      pNormalBufferSRV   = nullptr;
        ITextureView*      pMaterialBufferSRV = nullptr;
        ITextureView*      pMotionVectorsSRV  = nullptr;
        const HLSL::ScreenSpaceReflectionAttribs* pSSRAttribs = nullptr;
    };
    ScreenSpaceReflection(IRenderDevice* pDevice, const CreateInfo& CI);
    ~ScreenSpaceReflection();
    void          PrepareResources(IRenderDevice* pDevice, IDeviceContext* pDeviceContext, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags);
    void          Execute(const RenderAttributes& RenderAttribs);
    ITextureView* GetSSRRadianceSRV() const;

private:
    mifx_ssr* m_Impl = nullptr;
    Timer     m_FrameTimer;
};

class TemporalAntiAliasing
{
public:
    enum FEATURE_FLAGS : Uint32
    {
        FEATURE_FLAG_NONE               = 0u,
        FEATURE_FLAG_GAUSSIAN_WEIGHTING = 1u << 0u,
        FEATURE_FLAG_BICUBIC_FILTER     = 1u << 1u,
        FEATURE_FLAG_YCOCG_COLOR_SPACE  = 1u << 2u
    };
    struct CreateInfo
    {
        bool EnableAsyncCreation = false;
    };
    struct RenderAttributes
    {
        IRenderDevice*     pDevice         = nullptr;
        IRenderStateCache* pStateCache     = nullptr;
        IDeviceContext*    pDeviceContext  = nullptr;
        PostFXContext*     pPostFXContext  = nullptr;
        ITextureView*      pColorBufferSRV = nullptr;
        const HLSL::TemporalAntiAliasingAttribs* pTAAAttribs = nullptr;
        Uint32             AccumulationBufferIdx = 0;
    };
    static constexpr Uint32 kMaxAccumulationBuffers = 4; // one mifx_taa (= one history) per accumulation buffer index

    TemporalAntiAliasing(IRenderDevice* pDevice, const CreateInfo& CI);
    ~TemporalAntiAliasing();
    float2        GetJitterOffset(Uint32 AccumulationBufferIdx = 0) const;
    void          PrepareResources(IRenderDevice* pDevice, IDeviceContext* pDeviceContext, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags,
                                   Uint32 AccumulationBufferIdx = 0);
    void          Execute(const RenderAttributes& RenderAttribs);
    ITextureView* GetAccumulatedFrameSRV(bool IsPrevFrame = false, Uint32 AccumulationBufferIdx = 0) const;
    static float4x4 GetJitteredProjMatrix(const float4x4& Proj, const float2& Jitter);

private:
    struct Buffer
    {
        mifx_taa*                impl = nullptr;
        PostFXContext::FrameDesc frame;
    };
    Buffer m_Buffers[kMaxAccumulationBuffers];
};

class Bloom
{
public:
    enum FEATURE_FLAGS : Uint32
    {
        FEATURE_FLAG_NONE = 0u
    };
    struct CreateInfo
    {
        bool EnableAsyncCreation = false;
    };
    struct RenderAttributes
    {
        IRenderDevice*     pDevice         = nullptr;
        IRenderStateCache* pStateCache     = nullptr;
        IDeviceContext*    pDeviceContext  = nullptr;
        PostFXContext*     pPostFXContext  = nullptr;
        ITextureView*      pColorBufferSRV = nullptr;
        const HLSL::BloomAttribs* pBloomAttribs = nullptr;
    };
    Bloom(IRenderDevice* pDevice, const CreateInfo& CI);
    ~Bloom();
    void          PrepareResources(IRenderDevice* pDevice, IDeviceContext* pDeviceContext, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags);
    void          Execute(const RenderAttributes& RenderAttribs);
    ITextureView* GetBloomTextureSRV() const;

private:
    mifx_bloom* m_Impl = nullptr;
    Timer       m_FrameTimer;
};

class DepthOfField
{
public:
    enum FEATURE_FLAGS : Uint32
    {
        FEATURE_FLAG_NONE                      = 0u,
        FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING = 1u << 0u,
        FEATURE_FLAG_ENABLE_KARIS_INVERSE      = 1u << 1u
    };
    struct CreateInfo
    {
        bool EnableAsyncCreation = false;
    };
    struct RenderAttributes
    {
        IRenderDevice*     pDevice         = nullptr;
        IRenderStateCache* pStateCache     = nullptr;
        IDeviceContext*    pDeviceContext  = nullptr;
        PostFXContext*     pPostFXContext  = nullptr;
        ITextureView*      pColorBufferSRV = nullptr;
        ITextureView*      pDepthBufferSRV = nullptr;
        const HLSL::DepthOfFieldAttribs* pDOFAttribs = nullptr;
    };
    DepthOfField(IRenderDevice* pDevice, const CreateInfo& CI);
    ~DepthOfField();
    void          PrepareResources(IRenderDevice* pDevice, IDeviceContext* pDeviceContext, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags);
    void          Execute(const RenderAttributes& RenderAttribs);
    ITextureView* GetDepthOfFieldTextureSRV() const;

private:
    mifx_dof* m_Impl = nullptr;
    Timer     m_FrameTimer;
};

// The lighting half of RenderPBR.psh (GetSurfaceShadingInfo -> ApplyPunctualLight x N -> ApplyIBL -> ResolveLighting, RenderPBR.psh:299-359, 473-514) on a G-buffer with
// the USD targets (USD_Renderer.cpp:83-162), for a renderer that keeps PBR_Renderer's own constant blocks: the frame block is handed over as the bytes the renderer
// writes into its cbFrameAttribs buffer every frame (PBRFrameAttribs for the renderer's PBR_MAX_LIGHTS / PBR_MAX_SHADOW_MAPS, RenderPBR_Structures.fxh:11-24), the
// material block as PBRMaterialBasicAttribs (only its Workflow matters to a G-buffer shade), the IBL inputs as the SRVs PBR_Renderer hands out.
struct PBRGBufferShadeAttribs
{
    PostFXContext*                       pPostFXContext             = nullptr; // the HIP context / stream owner
    const void*                          pFrameAttribsData          = nullptr; // PBRFrameAttribs
    size_t                               FrameAttribsSize           = 0;
    Uint32                               MaxLightCount              = 16;      // PBR_Renderer::CreateInfo::MaxLightCount            (PBR_Renderer.hpp:245)
    Uint32                               MaxShadowCastingLightCount = 0;       // PBR_Renderer::CreateInfo::MaxShadowCastingLightCount (:248); 0 = ENABLE_SHADOWS off
    const HLSL::PBRMaterialBasicAttribs* pMaterial                  = nullptr; // nullptr = metallic-roughness
    ITextureView*                        pBaseColorSRV              = nullptr; // USD G-buffer targets
    ITextureView*                        pNormalSRV                 = nullptr;
    ITextureView*                        pMaterialDataSRV           = nullptr;
    ITextureView*                        pDepthSRV                  = nullptr;
    ITextureView*                        pBRDF_LUT_SRV              = nullptr; // PBR_Renderer::GetPreintegratedGGX_SRV
    ITextureView*                        pIrradianceCubeSRV         = nullptr; // GetIrradianceCubeSRV
    ITextureView*                        pPrefilteredEnvMapSRV      = nullptr; // GetPrefilteredEnvMapSRV
    const mifx_shadow_map_array*         pShadowMap                 = nullptr; // the shadow-map Texture2DArray as linear memory (MaxShadowCastingLightCount > 0)
    Uint32                               PCFKernelSize              = 3;       // PBR_Renderer::CreateInfo::PCFKernelSize (:227)
    float                                Background[4]              = {0, 0, 0, 0};
    ITextureView*                        pRadianceRTV               = nullptr; // SceneColor
    ITextureView*                        pSpecularIBLRTV            = nullptr; // the IBL target (may be nullptr)
};
void ShadeGBuffer(const PBRGBufferShadeAttribs& Attribs);

// The ToneMap() full-screen pass of the copy-frame shader (Hydrogent/shaders/HnCopyFrame.psh, HnPostProcessTask.cpp:974-1000) as a call
void ToneMapToTarget(PostFXContext& PostFX, ITextureView* pHDRColorSRV, const mifx_native_image& Target, const HLSL::ToneMappingAttribs& Attribs, float AverageLogLum,
                     bool ConvertOutputToSRGB);
} // namespace Diligent
