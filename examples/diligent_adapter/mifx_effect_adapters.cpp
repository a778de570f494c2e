// mifx_effect_adapters.cpp -- see mifx_effect_adapters.hpp.  Every method is a thin translation: texture views -> mifx_image2d through the
// application's interop (mifx_interop.hpp), attribute blocks copied byte for byte, status codes turned into the reference's logging.
#include "mifx_effect_adapters.hpp"

#include <algorithm>
#include <cstring>

namespace Diligent
{
// The HLSL attribute blocks and their mifx twins are the same bytes (mifx_sizeof() reports the library's view at run time)
static_assert(sizeof(HLSL::CameraAttribs) == sizeof(mifx_camera_attribs), "CameraAttribs");
static_assert(sizeof(HLSL::ScreenSpaceAmbientOcclusionAttribs) == sizeof(mifx_ssao_attribs), "ScreenSpaceAmbientOcclusionAttribs");
static_assert(sizeof(HLSL::ScreenSpaceReflectionAttribs) == sizeof(mifx_ssr_attribs), "ScreenSpaceReflectionAttribs");
static_assert(sizeof(HLSL::TemporalAntiAliasingAttribs) == sizeof(mifx_taa_attribs), "TemporalAntiAliasingAttribs");
static_assert(sizeof(HLSL::BloomAttribs) == sizeof(mifx_bloom_attribs), "BloomAttribs");
static_assert(sizeof(HLSL::DepthOfFieldAttribs) == sizeof(mifx_dof_attribs), "DepthOfFieldAttribs");
static_assert(sizeof(HLSL::ToneMappingAttribs) == sizeof(mifx_tone_mapping_attribs), "ToneMappingAttribs");

namespace
{
template <class Dst, class Src> Dst CopyBlock(const Src& s)
{
    static_assert(sizeof(Dst) == sizeof(Src), "attribute blocks must be byte-identical");
    Dst d;
    std::memcpy(&d, &s, sizeof(d));
    return d;
}
// the reference's methods return void: a failing call is logged with the library's detail string and the frame goes on (SURVEY 8b "Errors")
bool Succeeded(mifx_status st, const char* what)
{
    if (st >= 0) return true;
    LOG_ERROR_MESSAGE(what, ": ", mifx_status_string(st), " - ", mifx_last_error());
    return false;
}
// AlphaInterpolation as every UpdateConstantBuffer computes it (e.g. ScreenSpaceAmbientOcclusion.cpp:795): the effects' PSOs are "always ready"
// here, so the timer is never restarted after construction and the fade saturates after one second
float FrameAlpha(const Timer& t, float speed = 1.0f) { return std::min(std::max(t.GetElapsedTimef() * speed, 0.0f), 1.0f); }

ITextureView* OutputView(mifx_status st, const mifx_image2d& img) { return st >= 0 ? WrapMifxImage(img) : nullptr; }
} // namespace

// ---------------------------------------------------------------------------------------------------------------- PostFXContext
PostFXContext::PostFXContext(IRenderDevice*, const CreateInfo&, int HipDevice)
{
    mifx_device_desc        dev{HipDevice, /*hip_stream*/ nullptr};
    mifx_postfx_create_info info{NoiseBuffers::Sobol_256d, NoiseBuffers::ScramblingTile}; // the tables PostFXContext.cpp:155-190 uploads
    Succeeded(mifx_postfx_create(&dev, &info, &m_Mifx), "mifx_postfx_create");
}
PostFXContext::~PostFXContext() { mifx_postfx_destroy(m_Mifx); }

void PostFXContext::PrepareResources(IRenderDevice*, const FrameDesc& Desc, FEATURE_FLAGS FeatureFlags)
{
    m_FrameDesc    = Desc;
    m_FeatureFlags = FeatureFlags;
    if (m_Mifx == nullptr) return;
    const mifx_frame_desc frame{Desc.Index, Desc.Width, Desc.Height, Desc.OutputWidth, Desc.OutputHeight};
    Succeeded(mifx_postfx_prepare(m_Mifx, &frame, static_cast<uint32_t>(FeatureFlags)), "mifx_postfx_prepare");
}

void PostFXContext::Execute(const RenderAttributes& RenderAttribs)
{
    DEV_CHECK_ERR(RenderAttribs.pCurrCamera != nullptr && RenderAttribs.pPrevCamera != nullptr, "the camera blocks must be passed by pointer (pCameraAttribsCB lives in graphics memory)");
    if (m_Mifx == nullptr || RenderAttribs.pCurrCamera == nullptr || RenderAttribs.pPrevCamera == nullptr) return;
    Succeeded(mifx_postfx_set_stream(m_Mifx, GetMifxStream(RenderAttribs.pDeviceContext)), "mifx_postfx_set_stream");
    const mifx_image2d currDepth = GetMifxImage(RenderAttribs.pCurrDepthBufferSRV, MIFX_FORMAT_F32);
    const mifx_image2d prevDepth = GetMifxImage(RenderAttribs.pPrevDepthBufferSRV, MIFX_FORMAT_F32);
    const mifx_image2d motion    = GetMifxImage(RenderAttribs.pMotionVectorsSRV, MIFX_FORMAT_F32X2);
    const mifx_camera_attribs curr = CopyBlock<mifx_camera_attribs>(*RenderAttribs.pCurrCamera);
    const mifx_camera_attribs prev = CopyBlock<mifx_camera_attribs>(*RenderAttribs.pPrevCamera);
    const mifx_postfx_render_attribs ra{&currDepth, &prevDepth, &motion, &curr, &prev};
    Succeeded(mifx_postfx_execute(m_Mifx, &ra), "mifx_postfx_execute"); // blue noise, reprojected depth, closest motion; asynchronous on the stream
}

ITextureView* PostFXContext::Get2DBlueNoiseSRV(BLUE_NOISE_DIMENSION Dimension) const
{
    mifx_image2d img{};
    return OutputView(mifx_postfx_get_blue_noise(m_Mifx, static_cast<int32_t>(Dimension), &img), img);
}
ITextureView* PostFXContext::GetReprojectedDepth() const
{
    mifx_image2d img{};
    return OutputView(mifx_postfx_get_reprojected_depth(m_Mifx, &img), img);
}
ITextureView* PostFXContext::GetPreviousDepth() const
{
    mifx_image2d img{};
    return OutputView(mifx_postfx_get_previous_depth(m_Mifx, &img), img);
}
ITextureView* PostFXContext::GetClosestMotionVectors() const
{
    mifx_image2d img{};
    return OutputView(mifx_postfx_get_closest_motion(m_Mifx, &img), img);
}

PostFXContext::SupportedDeviceFeatures PostFXContext::GetSupportedFeatures() const
{
    SupportedDeviceFeatures        f;
    mifx_postfx_supported_features s{};
    if (m_Mifx != nullptr && Succeeded(mifx_postfx_get_supported_features(m_Mifx, &s), "mifx_postfx_get_supported_features"))
    {
        f.TransitionSubresources  = s.TransitionSubresources != 0;
        f.TextureSubresourceViews = s.TextureSubresourceViews != 0;
        f.CopyDepthToColor        = s.CopyDepthToColor != 0;
        f.ShaderBaseVertexOffset  = s.ShaderBaseVertexOffset != 0;
    }
    return f;
}
void PostFXContext::ClearRenderTarget(const TextureOperationAttribs& Attribs, ITextureView* pRTV, uint32_t Format, float ClearColor[])
{
    if (m_Mifx == nullptr) return;
    Succeeded(mifx_postfx_set_stream(m_Mifx, GetMifxStream(Attribs.pDeviceContext)), "mifx_postfx_set_stream");
    const mifx_image2d target = GetMifxImage(pRTV, Format);
    Succeeded(mifx_postfx_clear_render_target(m_Mifx, &target, ClearColor), "mifx_postfx_clear_render_target");
}
void PostFXContext::CopyTextureDepth(const TextureOperationAttribs& Attribs, ITextureView* pSRV, ITextureView* pRTV)
{
    if (m_Mifx == nullptr) return;
    Succeeded(mifx_postfx_set_stream(m_Mifx, GetMifxStream(Attribs.pDeviceContext)), "mifx_postfx_set_stream");
    const mifx_image2d src = GetMifxImage(pSRV, MIFX_FORMAT_F32), dst = GetMifxImage(pRTV, MIFX_FORMAT_F32);
    Succeeded(mifx_postfx_copy_texture_depth(m_Mifx, &src, &dst), "mifx_postfx_copy_texture_depth");
}
void PostFXContext::CopyTextureColor(const TextureOperationAttribs& Attribs, ITextureView* pSRV, ITextureView* pRTV)
{
    if (m_Mifx == nullptr) return;
    Succeeded(mifx_postfx_set_stream(m_Mifx, GetMifxStream(Attribs.pDeviceContext)), "mifx_postfx_set_stream");
    const mifx_image2d src = GetMifxImage(pSRV, MIFX_FORMAT_F32X4), dst = GetMifxImage(pRTV, MIFX_FORMAT_F32X4);
    Succeeded(mifx_postfx_copy_texture_color(m_Mifx, &src, &dst), "mifx_postfx_copy_texture_color");
}

// ---------------------------------------------------------------------------------------------------------------- ScreenSpaceAmbientOcclusion
ScreenSpaceAmbientOcclusion::ScreenSpaceAmbientOcclusion(IRenderDevice*, const CreateInfo&) { m_FrameTimer.Restart(); }
ScreenSpaceAmbientOcclusion::~ScreenSpaceAmbientOcclusion() { mifx_ssao_destroy(m_Impl); }

void ScreenSpaceAmbientOcclusion::PrepareResources(IRenderDevice*, IDeviceContext*, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags)
{
    mifx_postfx* ctx = pPostFXContext->GetMifxContext();
    if (m_Impl == nullptr && !Succeeded(mifx_ssao_create(ctx, &m_Impl), "mifx_ssao_create")) return;
    Succeeded(mifx_ssao_prepare(m_Impl, ctx, static_cast<uint32_t>(FeatureFlags)), "mifx_ssao_prepare"); // reversed depth follows the context, as in .cpp:72
}
void ScreenSpaceAmbientOcclusion::Execute(const RenderAttributes& RenderAttribs)
{
    DEV_CHECK_ERR(RenderAttribs.pPostFXContext != nullptr && RenderAttribs.pSSAOAttribs != nullptr, "RenderAttribs is incomplete");
    if (m_Impl == nullptr) return;
    const mifx_image2d depth  = GetMifxImage(RenderAttribs.pDepthBufferSRV, MIFX_FORMAT_F32);
    const mifx_image2d normal = GetMifxImage(RenderAttribs.pNormalBufferSRV, MIFX_FORMAT_F32X4);
    mifx_ssao_attribs  attribs = CopyBlock<mifx_ssao_attribs>(*RenderAttribs.pSSAOAttribs);
    attribs.AlphaInterpolation = FrameAlpha(m_FrameTimer); // history reset on skipped frames / first frame is the library's (mifx_ssao_execute -> MIFX_NO_HISTORY)
    const mifx_ssao_render_attribs ra{RenderAttribs.pPostFXContext->GetMifxContext(), &depth, &normal, &attribs};
    Succeeded(mifx_ssao_execute(m_Impl, &ra), "mifx_ssao_execute");
}
ITextureView* ScreenSpaceAmbientOcclusion::GetAmbientOcclusionSRV() const
{
    mifx_image2d img{};
    return OutputView(mifx_ssao_get_output(m_Impl, &img), img);
}

// ---------------------------------------------------------------------------------------------------------------- ScreenSpaceReflection
ScreenSpaceReflection::ScreenSpaceReflection(IRenderDevice*, const CreateInfo&) { m_FrameTimer.Restart(); }
ScreenSpaceReflection::~ScreenSpaceReflection() { mifx_ssr_destroy(m_Impl); }

void ScreenSpaceReflection::PrepareResources(IRenderDevice*, IDeviceContext*, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags)
{
    mifx_postfx* ctx = pPostFXContext->GetMifxContext();
    if (m_Impl == nullptr && !Succeeded(mifx_ssr_create(ctx, &m_Impl), "mifx_ssr_create")) return;
    Succeeded(mifx_ssr_prepare(m_Impl, ctx, static_cast<uint32_t>(FeatureFlags)), "mifx_ssr_prepare");
}
void ScreenSpaceReflection::Execute(const RenderAttributes& RenderAttribs)
{
    DEV_CHECK_ERR(RenderAttribs.pPostFXContext != nullptr && RenderAttribs.pSSRAttribs != nullptr, "RenderAttribs is incomplete");
    if (m_Impl == nullptr) return;
    const mifx_image2d color    = GetMifxImage(RenderAttribs.pColorBufferSRV, MIFX_FORMAT_F32X4);
    const mifx_image2d depth    = GetMifxImage(RenderAttribs.pDepthBufferSRV, MIFX_FORMAT_F32);
    const mifx_image2d normal   = GetMifxImage(RenderAttribs.pNormalBufferSRV, MIFX_FORMAT_F32X4);
    const mifx_image2d material = GetMifxImage(RenderAttribs.pMaterialBufferSRV, MIFX_FORMAT_F32X4);
    const mifx_image2d motion   = GetMifxImage(RenderAttribs.pMotionVectorsSRV, MIFX_FORMAT_F32X2);
    mifx_ssr_attribs   attribs  = CopyBlock<mifx_ssr_attribs>(*RenderAttribs.pSSRAttribs);
    attribs.AlphaInterpolation  = FrameAlpha(m_FrameTimer);
    const mifx_ssr_render_attribs ra{RenderAttribs.pPostFXContext->GetMifxContext(), &color, &depth, &normal, &material, &motion, &attribs};
    Succeeded(mifx_ssr_execute(m_Impl, &ra), "mifx_ssr_execute");
}
ITextureView* ScreenSpaceReflection::GetSSRRadianceSRV() const
{
    mifx_image2d img{};
    return OutputView(mifx_ssr_get_output(m_Impl, &img), img);
}

// ---------------------------------------------------------------------------------------------------------------- TemporalAntiAliasing
TemporalAntiAliasing::TemporalAntiAliasing(IRenderDevice*, const CreateInfo&) {}
TemporalAntiAliasing::~TemporalAntiAliasing()
{
    for (Buffer& b : m_Buffers) mifx_taa_destroy(b.impl);
}
float2 TemporalAntiAliasing::GetJitterOffset(Uint32 AccumulationBufferIdx) const // .cpp:63-78
{
    const Buffer& b = m_Buffers[AccumulationBufferIdx % kMaxAccumulationBuffers];
    float j[2] = {0.0f, 0.0f};
    if (b.frame.Width != 0 && b.frame.Height != 0) Succeeded(mifx_taa_get_jitter_offset(b.frame.Index, b.frame.Width, b.frame.Height, j), "mifx_taa_get_jitter_offset");
    return float2{j[0], j[1]};
}
float4x4 TemporalAntiAliasing::GetJitteredProjMatrix(const float4x4& Proj, const float2& Jitter) // .hpp:138-155
{
    float4x4    out = Proj;
    const float j[2] = {Jitter.x, Jitter.y};
    Succeeded(mifx_taa_get_jittered_proj_matrix(Proj.m, j, out.m), "mifx_taa_get_jittered_proj_matrix");
    return out;
}
void TemporalAntiAliasing::PrepareResources(IRenderDevice*, IDeviceContext*, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags, Uint32 AccumulationBufferIdx)
{
    DEV_CHECK_ERR(AccumulationBufferIdx < kMaxAccumulationBuffers, "accumulation buffer index out of range");
    Buffer&      b   = m_Buffers[AccumulationBufferIdx % kMaxAccumulationBuffers];
    mifx_postfx* ctx = pPostFXContext->GetMifxContext();
    b.frame          = pPostFXContext->GetFrameDesc();
    if (b.impl == nullptr && !Succeeded(mifx_taa_create(ctx, &b.impl), "mifx_taa_create")) return;
    Succeeded(mifx_taa_prepare(b.impl, ctx, static_cast<uint32_t>(FeatureFlags)), "mifx_taa_prepare");
}
void TemporalAntiAliasing::Execute(const RenderAttributes& RenderAttribs)
{
    DEV_CHECK_ERR(RenderAttribs.pPostFXContext != nullptr && RenderAttribs.pTAAAttribs != nullptr, "RenderAttribs is incomplete");
    Buffer& b = m_Buffers[RenderAttribs.AccumulationBufferIdx % kMaxAccumulationBuffers];
    if (b.impl == nullptr) return;
    const mifx_image2d     color   = GetMifxImage(RenderAttribs.pColorBufferSRV, MIFX_FORMAT_F32X4);
    const mifx_taa_attribs attribs = CopyBlock<mifx_taa_attribs>(*RenderAttribs.pTAAAttribs);
    const mifx_taa_render_attribs ra{RenderAttribs.pPostFXContext->GetMifxContext(), &color, &attribs};
    Succeeded(mifx_taa_execute(b.impl, &ra), "mifx_taa_execute"); // MIFX_NO_HISTORY (> 0) on the frames where the history was reset (.cpp:125-128)
}
ITextureView* TemporalAntiAliasing::GetAccumulatedFrameSRV(bool IsPrevFrame, Uint32 AccumulationBufferIdx) const
{
    mifx_image2d img{};
    return OutputView(mifx_taa_get_output(m_Buffers[AccumulationBufferIdx % kMaxAccumulationBuffers].impl, IsPrevFrame ? 1 : 0, &img), img);
}

// ---------------------------------------------------------------------------------------------------------------- Bloom
Bloom::Bloom(IRenderDevice*, const CreateInfo&) { m_FrameTimer.Restart(); }
Bloom::~Bloom() { mifx_bloom_destroy(m_Impl); }

void Bloom::PrepareResources(IRenderDevice*, IDeviceContext*, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags)
{
    mifx_postfx* ctx = pPostFXContext->GetMifxContext();
    if (m_Impl == nullptr && !Succeeded(mifx_bloom_create(ctx, &m_Impl), "mifx_bloom_create")) return;
    Succeeded(mifx_bloom_prepare(m_Impl, ctx, static_cast<uint32_t>(FeatureFlags)), "mifx_bloom_prepare");
}
void Bloom::Execute(const RenderAttributes& RenderAttribs)
{
    DEV_CHECK_ERR(RenderAttribs.pPostFXContext != nullptr && RenderAttribs.pBloomAttribs != nullptr, "RenderAttribs is incomplete");
    if (m_Impl == nullptr) return;
    const mifx_image2d color   = GetMifxImage(RenderAttribs.pColorBufferSRV, MIFX_FORMAT_F32X4);
    mifx_bloom_attribs attribs = CopyBlock<mifx_bloom_attribs>(*RenderAttribs.pBloomAttribs);
    attribs.AlphaInterpolation = FrameAlpha(m_FrameTimer); // Bloom.cpp:277
    const mifx_bloom_render_attribs ra{RenderAttribs.pPostFXContext->GetMifxContext(), &color, &attribs};
    Succeeded(mifx_bloom_execute(m_Impl, &ra), "mifx_bloom_execute");
}
ITextureView* Bloom::GetBloomTextureSRV() const
{
    mifx_image2d img{};
    return OutputView(mifx_bloom_get_output(m_Impl, &img), img);
}

// ---------------------------------------------------------------------------------------------------------------- DepthOfField
DepthOfField::DepthOfField(IRenderDevice*, const CreateInfo&) { m_FrameTimer.Restart(); }
DepthOfField::~DepthOfField() { mifx_dof_destroy(m_Impl); }

void DepthOfField::PrepareResources(IRenderDevice*, IDeviceContext*, PostFXContext* pPostFXContext, FEATURE_FLAGS FeatureFlags)
{
    mifx_postfx* ctx = pPostFXContext->GetMifxContext();
    if (m_Impl == nullptr && !Succeeded(mifx_dof_create(ctx, &m_Impl), "mifx_dof_create")) return;
    Succeeded(mifx_dof_prepare(m_Impl, ctx, static_cast<uint32_t>(FeatureFlags)), "mifx_dof_prepare");
}
void DepthOfField::Execute(const RenderAttributes& RenderAttribs)
{
    DEV_CHECK_ERR(RenderAttribs.pPostFXContext != nullptr && RenderAttribs.pDOFAttribs != nullptr, "RenderAttribs is incomplete");
    if (m_Impl == nullptr) return;
    const mifx_image2d color   = GetMifxImage(RenderAttribs.pColorBufferSRV, MIFX_FORMAT_F32X4);
    const mifx_image2d depth   = GetMifxImage(RenderAttribs.pDepthBufferSRV, MIFX_FORMAT_F32);
    mifx_dof_attribs   attribs = CopyBlock<mifx_dof_attribs>(*RenderAttribs.pDOFAttribs);
    attribs.AlphaInterpolation = FrameAlpha(m_FrameTimer, RenderAttribs.pPostFXContext->GetInterpolationSpeed()); // DepthOfField.cpp:797
    const mifx_dof_render_attribs ra{RenderAttribs.pPostFXContext->GetMifxContext(), &color, &depth, &attribs};
    Succeeded(mifx_dof_execute(m_Impl, &ra), "mifx_dof_execute"); // the camera and the closest motion vectors come from the context, as in the reference
}
ITextureView* DepthOfField::GetDepthOfFieldTextureSRV() const
{
    mifx_image2d img{};
    return OutputView(mifx_dof_get_output(m_Impl, &img), img);
}

// ---------------------------------------------------------------------------------------------------------------- PBR shade of a G-buffer
static_assert(sizeof(HLSL::PBRMaterialBasicAttribs) == sizeof(mifx_pbr_material_basic_attribs), "PBRMaterialBasicAttribs");
static_assert(sizeof(HLSL::PBRRendererShaderParameters) == sizeof(mifx_pbr_renderer_shader_parameters), "PBRRendererShaderParameters");
static_assert(sizeof(HLSL::PBRLightAttribs) == sizeof(mifx_pbr_light_attribs), "PBRLightAttribs");
static_assert(sizeof(HLSL::PBRShadowMapInfo) == sizeof(mifx_pbr_shadow_map_info), "PBRShadowMapInfo");

void ShadeGBuffer(const PBRGBufferShadeAttribs& A)
{
    DEV_CHECK_ERR(A.pPostFXContext != nullptr && A.pFrameAttribsData != nullptr, "PBRGBufferShadeAttribs is incomplete");
    if (A.pPostFXContext == nullptr || A.pPostFXContext->GetMifxContext() == nullptr) return;
    const mifx_image2d bc = GetMifxImage(A.pBaseColorSRV, MIFX_FORMAT_F32X4), nrm = GetMifxImage(A.pNormalSRV, MIFX_FORMAT_F32X4), mat = GetMifxImage(A.pMaterialDataSRV, MIFX_FORMAT_F32X4),
                       depth = GetMifxImage(A.pDepthSRV, MIFX_FORMAT_F32), lut = GetMifxImage(A.pBRDF_LUT_SRV, MIFX_FORMAT_F32X2), rad = GetMifxImage(A.pRadianceRTV, MIFX_FORMAT_F32X4),
                       spec = GetMifxImage(A.pSpecularIBLRTV, MIFX_FORMAT_F32X4);
    mifx_cubemap irradiance{}, prefiltered{};
    if (!GetMifxCubemap(A.pIrradianceCubeSRV, irradiance) || !GetMifxCubemap(A.pPrefilteredEnvMapSRV, prefiltered))
    {
        LOG_ERROR_MESSAGE("ShadeGBuffer: the IBL cube maps are not shared with HIP");
        return;
    }
    const mifx_gbuffer gbuffer{&bc, &nrm, &mat, &depth, nullptr, nullptr};
    const mifx_ibl     ibl{&lut, &irradiance, &prefiltered};
    mifx_pbr_material_basic_attribs material{};
    if (A.pMaterial != nullptr) material = CopyBlock<mifx_pbr_material_basic_attribs>(*A.pMaterial);
    Succeeded(mifx_pbr_shade_execute_frame_attribs(A.pPostFXContext->GetMifxContext(), &gbuffer, A.pFrameAttribsData, A.FrameAttribsSize, A.MaxLightCount, A.MaxShadowCastingLightCount,
                                                   A.pMaterial != nullptr ? &material : nullptr, &ibl, A.pShadowMap, A.PCFKernelSize, A.Background, &rad,
                                                   A.pSpecularIBLRTV != nullptr ? &spec : nullptr),
              "mifx_pbr_shade_execute_frame_attribs");
}

// ---------------------------------------------------------------------------------------------------------------- tone map of the copy-frame pass
void ToneMapToTarget(PostFXContext& PostFX, ITextureView* pHDRColorSRV, const mifx_native_image& Target, const HLSL::ToneMappingAttribs& Attribs, float AverageLogLum,
                     bool ConvertOutputToSRGB)
{
    const mifx_image2d              hdr = GetMifxImage(pHDRColorSRV, MIFX_FORMAT_F32X4);
    const mifx_tone_mapping_attribs tm  = CopyBlock<mifx_tone_mapping_attribs>(Attribs);
    Succeeded(mifx_tonemap_execute_native(PostFX.GetMifxContext(), &hdr, &Target, &tm, AverageLogLum, ConvertOutputToSRGB ? uint32_t(MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB) : 0u),
              "mifx_tonemap_execute_native");
}
} // namespace Diligent
