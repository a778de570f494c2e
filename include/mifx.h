/* mifx.h -- C ABI of the MI355X-native DiligentFX hot path (PBR shade + PostProcess chain).
 *
 * This is the drop-in boundary: every entry point replaces one C++ interface of the reference
 * (cited file:line, relative to the DiligentFX tree) without the DiligentCore render-device abstraction.
 * Conventions (same as the reference's protocol):
 *   - images are pitched HBM arrays of float / float2 / float4 ("fp32-storage mode"); inputs are borrowed
 *     for the duration of the call, outputs are owned by the effect object and stay valid until the next
 *     prepare() that changes size or feature flags;
 *   - attribs structs are byte-identical to the reference HLSL/C++ structs;
 *   - one object <-> one HIP stream, externally synchronised; execute() is asynchronous on that stream;
 *   - per frame: mifx_postfx_prepare -> <effect>_prepare ... -> mifx_postfx_execute -> <effect>_execute ...
 *   - no hidden state: frame index, TAA jitter and AlphaInterpolation are explicit inputs
 *     (the reference derives AlphaInterpolation from a wall clock, ScreenSpaceAmbientOcclusion.cpp:790-795);
 *   - errors are returned (status < 0), never asserted; status > 0 is informational.
 */
#ifndef MIFX_H
#define MIFX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIFX_API __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------------ status */
typedef int32_t mifx_status;
enum
{
    MIFX_OK                  = 0,
    MIFX_NO_HISTORY          = 1,  /* informational: temporal history was (re)initialised this frame */
    MIFX_ERR_INVALID_ARG     = -1,
    MIFX_ERR_INVALID_OP      = -2, /* e.g. execute() before prepare() (TemporalAntiAliasing.cpp:178-184) */
    MIFX_ERR_HIP             = -3,
    MIFX_ERR_OUT_OF_MEMORY   = -4,
    MIFX_ERR_NOT_IMPLEMENTED = -5,
    MIFX_ERR_COMM            = -6
};
MIFX_API const char* mifx_status_string(mifx_status s);
MIFX_API const char* mifx_last_error(void); /* thread-local detail of the last failure */
/* rocTX ranges around the effects and their passes, named after the reference's ScopedDebugGroup markers ("ScreenSpaceAmbientOcclusion" > "ComputeAmbientOcclusion", ...;
 * ScreenSpaceAmbientOcclusion.cpp:363,976): visible in `rocprofv3 --marker-trace`. Off by default; also switched on by the environment variable MIFX_ROCTX=1. */
MIFX_API void mifx_set_markers(int32_t enable);

/* ------------------------------------------------------------------------------------------------ images */
enum
{
    MIFX_FORMAT_F32   = 1, /* 1 x float  (depth, AO, roughness, variance)  */
    MIFX_FORMAT_F32X2 = 2, /* 2 x float  (motion, blue noise)              */
    MIFX_FORMAT_F32X4 = 4, /* 4 x float  (colour, normal, material, ...)   */
    MIFX_FORMAT_F16X4 = 8, /* 4 x binary16: the RGBA16_FLOAT of the reference's colour targets. The 4-channel texel of the native-storage build of this
                              library (libmifx_h4.so, mifx_storage_mode() == MIFX_STORAGE_RGBA16F) wherever the fp32 build takes MIFX_FORMAT_F32X4. */
    /* the narrow formats of the reference's intermediate targets; in the native-storage build the planes an effect hands out have them (fp32 build: F32 / F32X2 / F32X4):
     * ambient occlusion and SSR roughness R8_UNORM, SSAO history length / SSR variance / SSR resolved depth R16_FLOAT, closest motion RG16_FLOAT, Bloom R11G11B10_FLOAT */
    MIFX_FORMAT_U8        = 16,
    MIFX_FORMAT_F16       = 32,
    MIFX_FORMAT_F16X2     = 64,
    MIFX_FORMAT_R11G11B10 = 128,
    MIFX_FORMAT_U16       = 256 /* R16_UNORM: the dilated / blurred circle of confusion of depth of field (DepthOfField.cpp:227-229) */
};
/* Which texels the images have -- inputs borrowed from the caller, effect-owned planes, outputs: a property of the library build, fixed when the
 * application picks the library (the same C ABI, two shared objects):
 *   libmifx.so     MIFX_STORAGE_FP32     float4 / float2 / float planes, the parity contract of BASELINE.json ("pitched float4/float HBM arrays")
 *   libmifx_h4.so  MIFX_STORAGE_RGBA16F  the reference's own target formats ("native storage"); the arithmetic stays fp32 as on the GPU path of the reference, a value
 *                                        takes the rounding of its target's format when it is stored:
 *        every 4-channel image           RGBA16_FLOAT (MIFX_FORMAT_F16X4): SceneColor / Normal / IBL of the G-buffer, the SSR ray / resolve / history / output targets,
 *                                        the TAA accumulation buffers (HnBeginFrameTask.cpp:63-69, ScreenSpaceReflection.cpp:203-290, TemporalAntiAliasing.cpp:107);
 *                                        BaseColor RGBA8 and Material RG8 of the G-buffer are narrower in the reference (mifx_*_native entry points read those)
 *        ambient occlusion (all stages), SSR roughness            R8_UNORM   (MIFX_FORMAT_U8;   ScreenSpaceAmbientOcclusion.hpp:255, ScreenSpaceReflection.cpp:155)
 *        SSAO history length, SSR variance / resolved depth       R16_FLOAT  (MIFX_FORMAT_F16;  ScreenSpaceAmbientOcclusion.hpp:256, ScreenSpaceReflection.cpp:236-275)
 *        closest motion                                           RG16_FLOAT (MIFX_FORMAT_F16X2; PostFXContext.cpp:281)
 *        Bloom levels and Bloom's output target                   R11G11B10_FLOAT (MIFX_FORMAT_R11G11B10; Bloom.cpp:111-137): mifx_bloom_get_output hands out a 4-byte plane, which
 *                                                                 mifx_tonemap_execute / _native / _auto and mifx_autoexposure_execute take beside the RGBA16_FLOAT frame
 *        depth of field: circle of confusion and its history      R16_FLOAT  (MIFX_FORMAT_F16;  DepthOfField.cpp:196-223)
 *                        dilated / blurred circle of confusion    R16_UNORM  (MIFX_FORMAT_U16;  DepthOfField.cpp:227-253: the format the reference takes where the device
 *                                                                 supports it, and every device it targets does); the separated circle of confusion, which the kernels
 *                                                                 read through the signed one, takes the same rounding
 *                        the combined output (DepthOfField.cpp:281-289, R11G11B10_FLOAT) like Bloom's output: those values, alpha 1, in an RGBA16_FLOAT plane
 *        depth, the depth pyramids, reprojected depth, the reflection mask, the motion input, cube maps and the LUT    fp32 in both builds; with the HALF_PRECISION_DEPTH flags of
 *                                                                 PostFX / SSAO the native-storage build rounds the reprojected / previous depth and SSAO's two depth pyramids to
 *                                                                 R16_UNORM values as the reference's targets do (PostFXContext.cpp:259-270, ScreenSpaceAmbientOcclusion.cpp:95-97) */
enum
{
    MIFX_STORAGE_FP32    = 0,
    MIFX_STORAGE_RGBA16F = 1
};
MIFX_API uint32_t mifx_storage_mode(void);
typedef struct mifx_image2d
{
    void*    data;        /* device pointer (HIP) */
    uint32_t width;
    uint32_t height;
    uint32_t pitch_bytes; /* row pitch, multiple of the texel size; pitch_bytes * height < 4 GiB (32-bit texel offsets inside the kernels) */
    uint32_t format;      /* MIFX_FORMAT_* */
} mifx_image2d;

/* Cube map, D3D face order +X,-X,+Y,-Y,+Z,-Z; faces of one mip are stacked vertically (6*size rows), float4 texels. */
typedef struct mifx_cubemap
{
    const void* mip_data[16]; /* device pointers, mip_data[k] has (size>>k) x 6*(size>>k) float4 texels, tightly packed */
    uint32_t    size;
    uint32_t    mip_count;
} mifx_cubemap;

/* Equirectangular ("sphere") environment map: a float4 Texture2D with its mip chain (ENV_MAP_TYPE_SPHERE of the reference: looked up at
 * TransformDirectionToSphereMapUV(direction), Shaders/Common/public/ShaderUtilities.fxh:98-102). */
typedef struct mifx_spheremap
{
    const void* mip_data[16]; /* device pointers, mip_data[k] has max(width>>k, 1) x max(height>>k, 1) float4 texels, tightly packed */
    uint32_t    width, height;
    uint32_t    mip_count;
} mifx_spheremap;

/* ------------------------------------------------------------------------------------------------ structs shared with the reference (byte-identical) */

/* CameraAttribs -- Shaders/Common/public/BasicStructures.fxh:84-149 (576 bytes). Matrices are row-major, row-vector
 * convention (clip = mul(float4(p,1), M)). */
typedef struct mifx_camera_attribs
{
    float    f4Position[4];
    float    f4ViewportSize[4]; /* (width, height, 1/width, 1/height) */
    float    fNearPlaneZ, fFarPlaneZ, fNearPlaneDepth, fFarPlaneDepth;
    float    fSceneNearZ, fSceneFarZ, fSceneNearDepth, fSceneFarDepth;
    float    fHandness;
    uint32_t uiFrameIndex;
    float    Padding0, Padding1;
    float    fFocusDistance, fFStop, fFocalLength, fSensorWidth;
    float    fSensorHeight, fExposure;
    float    f2Jitter[2];
    float    mView[16], mProj[16], mViewProj[16], mViewInv[16], mProjInv[16], mViewProjInv[16];
    float    f4ExtraData[5][4];
} mifx_camera_attribs;

/* ToneMappingAttribs -- Shaders/PostProcess/ToneMapping/public/ToneMappingStructures.fxh:24-52 (48 bytes) */
enum
{
    MIFX_TONE_MAPPING_MODE_NONE = 0, MIFX_TONE_MAPPING_MODE_EXP, MIFX_TONE_MAPPING_MODE_REINHARD, MIFX_TONE_MAPPING_MODE_REINHARD_MOD,
    MIFX_TONE_MAPPING_MODE_UNCHARTED2, MIFX_TONE_MAPPING_MODE_FILMIC_ALU, MIFX_TONE_MAPPING_MODE_LOGARITHMIC,
    MIFX_TONE_MAPPING_MODE_ADAPTIVE_LOG, MIFX_TONE_MAPPING_MODE_AGX, MIFX_TONE_MAPPING_MODE_AGX_CUSTOM,
    MIFX_TONE_MAPPING_MODE_PBR_NEUTRAL, MIFX_TONE_MAPPING_MODE_COMMERCE
};
typedef struct mifx_tone_mapping_attribs
{
    int32_t  iToneMappingMode; /* default UNCHARTED2 */
    int32_t  bAutoExposure;
    float    fMiddleGray;      /* 0.18 */
    int32_t  bLightAdaptation;
    float    fWhitePoint;      /* 3.0 */
    float    fLuminanceSaturation; /* 1.0 */
    uint32_t Padding0, Padding1;
    float    AgXSaturation, AgXSlope, AgXPower, AgXOffset; /* AgXAttribs, :24-30; defaults 1,1,1,0 */
} mifx_tone_mapping_attribs;

/* ScreenSpaceAmbientOcclusionAttribs -- .../ScreenSpaceAmbientOcclusionStructures.fxh:64-98 (48 bytes) */
enum { MIFX_SSAO_ALGORITHM_GTAO = 0, MIFX_SSAO_ALGORITHM_HBAO = 1, MIFX_SSAO_ALGORITHM_VBAO = 2 };
typedef struct mifx_ssao_attribs
{
    float    EffectRadius;                /* 1.0   */
    float    EffectFalloffRange;          /* 0.615 */
    float    RadiusMultiplier;            /* 1.457 */
    float    DepthMIPSamplingOffset;      /* 3.3   */
    float    TemporalStabilityFactor;     /* 0.9   */
    float    SpatialReconstructionRadius; /* 4.0   */
    int32_t  ResetAccumulation;           /* 0; OR-ed with the frame-index rule (ScreenSpaceAmbientOcclusion.cpp:797-800) */
    float    AlphaInterpolation;          /* 1.0; explicit here, wall-clock in the reference */
    float    BitmaskThickness;            /* 0.5   */
    uint32_t Algorithm;                   /* GTAO  */
    float    Padding0, Padding1;
} mifx_ssao_attribs;

/* ScreenSpaceReflectionAttribs -- .../ScreenSpaceReflectionStructures.fxh:43-80 (48 bytes) */
typedef struct mifx_ssr_attribs
{
    float    DepthBufferThickness;               /* 0.025 */
    float    RoughnessThreshold;                 /* 0.2   */
    uint32_t MostDetailedMip;                    /* 0     */
    int32_t  IsRoughnessPerceptual;              /* 1     */
    uint32_t RoughnessChannel;                   /* 0     */
    uint32_t MaxTraversalIntersections;          /* 128   */
    float    GGXImportanceSampleBias;            /* 0.3   */
    float    SpatialReconstructionRadius;        /* 4.0   */
    float    TemporalRadianceStabilityFactor;    /* 1.0   */
    float    TemporalVarianceStabilityFactor;    /* 0.9   */
    float    BilateralCleanupSpatialSigmaFactor; /* 0.9   */
    float    AlphaInterpolation;                 /* 1.0   */
} mifx_ssr_attribs;

/* BloomAttribs -- Shaders/PostProcess/Bloom/public/BloomStructures.fxh:12-34 (32 bytes) */
typedef struct mifx_bloom_attribs
{
    float Intensity;          /* 0.15  */
    float Threshold;          /* 1.0   */
    float SoftTreshold;       /* 0.125 */
    float Radius;             /* 0.75  */
    float AlphaInterpolation; /* 1.0   */
    float Padding0, Padding1, Padding2;
} mifx_bloom_attribs;

/* DepthOfFieldAttribs -- Shaders/PostProcess/DepthOfField/public/DepthOfFieldStructures.fxh:31-56 (32 bytes) */
typedef struct mifx_dof_attribs
{
    float   MaxCircleOfConfusion;    /* 0.01   */
    float   TemporalStabilityFactor; /* 0.9375 */
    int32_t BokehKernelRingCount;    /* 5      */
    int32_t BokehKernelRingDensity;  /* 7      */
    float   AlphaInterpolation;      /* 1.0; explicit here, wall-clock in the reference (DepthOfField.cpp:797) */
    float   Padding0, Padding1, Padding2;
} mifx_dof_attribs;

/* TemporalAntiAliasingAttribs -- .../TemporalAntiAliasingStructures.fxh:35-46 (16 bytes) */
typedef struct mifx_taa_attribs
{
    float   TemporalStabilityFactor; /* 0.9375 */
    int32_t ResetAccumulation;
    int32_t SkipRejection;
    float   Padding0;
} mifx_taa_attribs;

/* PBRLightAttribs -- Shaders/PBR/public/PBR_Structures.fxh:309-330 (64 bytes) */
enum { MIFX_PBR_LIGHT_TYPE_DIRECTIONAL = 1, MIFX_PBR_LIGHT_TYPE_POINT = 2, MIFX_PBR_LIGHT_TYPE_SPOT = 3 };
typedef struct mifx_pbr_light_attribs
{
    int32_t Type;
    float   PosX, PosY, PosZ;
    float   DirectionX, DirectionY, DirectionZ;
    int32_t ShadowMapIndex; /* -1, or an index into mifx_pbr_shadows::shadow_maps (mifx_pbr_shade_execute_with_shadows only) */
    float   IntensityR, IntensityG, IntensityB;
    float   Range4;
    float   SpotAngleScale, SpotAngleOffset;
    float   Padding0, Padding1;
} mifx_pbr_light_attribs;

#define MIFX_PBR_MAX_LIGHTS 16 /* PBR/interface/PBR_Renderer.hpp:245 */

/* The lighting-relevant subset of PBRRendererShaderParameters (PBR_Structures.fxh:126-149) + the light list of
 * PBRFrameAttribs (Shaders/PBR/private/RenderPBR_Structures.fxh:11-24): what the kernels take.  A renderer that holds the reference's own blocks passes them as they
 * are to mifx_pbr_shade_execute_frame_attribs (below), which fills this struct from them. */
typedef struct mifx_pbr_shade_attribs
{
    float                  IBLScale[4];            /* Renderer.IBLScale                  */
    float                  OcclusionStrength;      /* Renderer.OcclusionStrength         */
    float                  EmissionScale;          /* Renderer.EmissionScale             */
    float                  PrefilteredCubeLastMip; /* Renderer.PrefilteredCubeLastMip    */
    int32_t                LightCount;             /* Renderer.LightCount, <= MIFX_PBR_MAX_LIGHTS */
    mifx_pbr_light_attribs Lights[MIFX_PBR_MAX_LIGHTS];
    int32_t                Workflow;               /* PBRMaterialBasicAttribs::Workflow of the frame's G-buffer (PBR_Structures.fxh:25-31): MIFX_PBR_WORKFLOW_* */
    int32_t                Padding[3];
} mifx_pbr_shade_attribs;
/* How the material plane of the G-buffer is read (ReadBaseLayerProperties, RenderPBR.psh:151-173 -> GetSurfaceReflectance, PBR_Shading.fxh:376-426):
 *   METALLIC_ROUGHNESS  (default): material = (PerceptualRoughness, Metallic, -, -), the USD G-buffer contract (USD_Renderer.cpp:98);
 *   SPECULAR_GLOSSINESS: material = PhysicalDesc as fetched: rgb = specular colour in sRGB space (FastSRGBToLinear is applied, :151-158), a = glossiness. */
enum
{
    MIFX_PBR_WORKFLOW_METALLIC_ROUGHNESS  = 0,
    MIFX_PBR_WORKFLOW_SPECULAR_GLOSSINESS = 1
};

/* ------------------------------------------------------------------------------------------------ context / PostFXContext */
typedef struct mifx_device_desc
{
    int32_t device;     /* HIP device ordinal */
    void*   hip_stream; /* hipStream_t the objects created from this context record on (NULL = default stream) */
} mifx_device_desc;

typedef struct mifx_postfx mifx_postfx; /* == PostFXContext, PostProcess/Common/interface/PostFXContext.hpp:48-263 */

/* PostFXContext::FrameDesc, PostFXContext.hpp:74-91 */
typedef struct mifx_frame_desc
{
    uint32_t Index;
    uint32_t Width, Height;
    uint32_t OutputWidth, OutputHeight;
} mifx_frame_desc;

enum
{
    MIFX_POSTFX_FEATURE_FLAG_NONE               = 0,
    MIFX_POSTFX_FEATURE_FLAG_REVERSED_DEPTH     = 1 << 0, /* PostFXContext.hpp:55: near plane = depth 1, background = depth 0; SSAO / SSR follow it (…AmbientOcclusion.cpp:72, …Reflection.cpp:73) */
    MIFX_POSTFX_FEATURE_FLAG_HALF_PRECISION_DEPTH = 1 << 1, /* PostFXContext.cpp:259-270: the reprojected and the previous depth are R16_UNORM targets.  fp32 build: full precision
                                                             * like every plane; native-storage build: the two hold the values R16_UNORM targets keep (in 4-byte texels: the flag
                                                             * is a run-time switch, a plane's texel type a compile-time one), decided like the reference's formats when the
                                                             * planes are created, i.e. on a change of the frame size (:246-247) */
    MIFX_POSTFX_FEATURE_FLAG_TEMPORAL_UPSCALING = 1 << 2  /* PostFXContext.hpp:56: passes that run after the temporal up-scaler work at FrameDesc.OutputWidth x OutputHeight --
                                                             here Bloom, which sizes its pyramid and output from the output size (Bloom.cpp:84-85); the passes before it
                                                             (prep, SSAO, SSR, TAA) keep FrameDesc.Width x Height. Requires OutputWidth / OutputHeight != 0. */
};

/* The Sobol sequence / scrambling tile tables of the blue-noise sampler. The reference keeps them in
 * PostProcess/Common/src/SamplerBlueNoiseErrorDistribution_128x128_OptimizedFor_2d2d2d2d_1spp.cpp (Sobol_256d[256],
 * ScramblingTile[128*128*8]) and uploads them in the PostFXContext ctor (PostFXContext.cpp:152-190); here the caller
 * (the adapter that links DiligentFX, or a test fixture) hands them over. Host pointers, copied during create.
 * NULL tables: blue noise is unavailable and SSAO/SSR execute return MIFX_ERR_INVALID_OP. */
typedef struct mifx_postfx_create_info
{
    const uint8_t* sobol_256d;      /* 256 bytes          */
    const uint8_t* scrambling_tile; /* 128*128*8 bytes    */
} mifx_postfx_create_info;
MIFX_API mifx_status mifx_postfx_create(const mifx_device_desc* dev, const mifx_postfx_create_info* info, mifx_postfx** out); /* PostFXContext ctor, PostFXContext.cpp:139 */
MIFX_API void        mifx_postfx_destroy(mifx_postfx* ctx);
MIFX_API mifx_status mifx_postfx_set_stream(mifx_postfx* ctx, void* hip_stream);
MIFX_API mifx_status mifx_postfx_prepare(mifx_postfx* ctx, const mifx_frame_desc* frame, uint32_t feature_flags); /* PrepareResources, PostFXContext.cpp:241 */

typedef struct mifx_postfx_render_attribs /* PostFXContext::RenderAttributes, PostFXContext.hpp:93-120 */
{
    const mifx_image2d*        curr_depth;  /* F32   */
    const mifx_image2d*        prev_depth;  /* F32   */
    const mifx_image2d*        motion;      /* F32X2, NDC units */
    const mifx_camera_attribs* curr_camera;
    const mifx_camera_attribs* prev_camera;
} mifx_postfx_render_attribs;
MIFX_API mifx_status mifx_postfx_execute(mifx_postfx* ctx, const mifx_postfx_render_attribs* attribs); /* PostFXContext::Execute, PostFXContext.cpp:287 */
MIFX_API mifx_status mifx_postfx_get_reprojected_depth(mifx_postfx* ctx, mifx_image2d* out);    /* GetReprojectedDepth, PostFXContext.hpp:143 */
MIFX_API mifx_status mifx_postfx_get_previous_depth(mifx_postfx* ctx, mifx_image2d* out);       /* GetPreviousDepth */
MIFX_API mifx_status mifx_postfx_get_closest_motion(mifx_postfx* ctx, mifx_image2d* out);       /* GetClosestMotionVectors */
MIFX_API mifx_status mifx_postfx_get_blue_noise(mifx_postfx* ctx, int32_t dimension, mifx_image2d* out); /* Get2DBlueNoiseSRV(XY=0 / ZW=1) */
/* PostFXContext::SupportedDeviceFeatures / GetSupportedFeatures (PostFXContext.hpp:114-120, 153): device capabilities the reference's effects branch on. All four
 * hold by construction here (every pass addresses any mip level of any plane and takes its frame / mip index as an argument). */
typedef struct mifx_postfx_supported_features
{
    int32_t TransitionSubresources;
    int32_t TextureSubresourceViews;
    int32_t CopyDepthToColor;
    int32_t ShaderBaseVertexOffset;
} mifx_postfx_supported_features;
MIFX_API mifx_status mifx_postfx_get_supported_features(mifx_postfx* ctx, mifx_postfx_supported_features* out);
/* PostFXContext::ClearRenderTarget (PostFXContext.hpp:168, .cpp:370-376): every texel of a float plane := clear_color (one value per channel), on the context's stream. */
MIFX_API mifx_status mifx_postfx_clear_render_target(mifx_postfx* ctx, const mifx_image2d* target, const float clear_color[4]);
/* PostFXContext::CopyTextureDepth / CopyTextureColor (PostFXContext.hpp:170-172, .cpp:378-438): the reference's full-screen copy draws; every caller copies between targets
 * of one size, where its point / linear CLAMP samplers return the texel itself. Source and target must have the same size and format (MIFX_ERR_INVALID_ARG otherwise). */
MIFX_API mifx_status mifx_postfx_copy_texture_depth(mifx_postfx* ctx, const mifx_image2d* src, const mifx_image2d* dst);
MIFX_API mifx_status mifx_postfx_copy_texture_color(mifx_postfx* ctx, const mifx_image2d* src, const mifx_image2d* dst);

/* ------------------------------------------------------------------------------------------------ ToneMapping (the "ToneMapping::Execute" of north_star) */
enum { MIFX_TONEMAP_FLAG_NONE = 0, MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB = 1 };
/* Full-screen application of ToneMap() (ToneMapping.fxh:87-226) as in Hydrogent/shaders/HnCopyFrame.psh:27-36,61-63. */
MIFX_API mifx_status mifx_tonemap_execute(mifx_postfx* ctx, const mifx_image2d* hdr_in, const mifx_image2d* ldr_out,
                                          const mifx_tone_mapping_attribs* attribs, float ave_log_lum, uint32_t flags);
/* Host helper == ToneMapping::ReverseExpToneMap... kept minimal: Components/src/ToneMapping.cpp:43-83 */
MIFX_API mifx_status mifx_reverse_exp_tone_map(const float ldr_rgb[3], float middle_gray, float ave_log_lum, float out_hdr_rgb[3]);

/* ------------------------------------------------------------------------------------------------ ScreenSpaceAmbientOcclusion */
typedef struct mifx_ssao mifx_ssao; /* ScreenSpaceAmbientOcclusion.hpp:57-262 */
enum
{
    MIFX_SSAO_FEATURE_FLAG_NONE            = 0,
    MIFX_SSAO_FEATURE_FLAG_HALF_PRECISION_DEPTH = 1 << 0, /* self-occlusion offset 5e-3 (SSAO_ComputeAmbientOcclusion.fx:145-150); the prefiltered and the convoluted depth pyramid are R16_UNORM
                                                           * targets in the reference (.cpp:95-97): the native-storage build rounds their values accordingly (see the PostFX flag) */
    MIFX_SSAO_FEATURE_FLAG_HALF_RESOLUTION = 1 << 1       /* checkerboard depth (A1), pyramid + AO at half size, bilateral upsampling (A4) */
}; /* == ScreenSpaceAmbientOcclusion::FEATURE_FLAGS (ScreenSpaceAmbientOcclusion.hpp:59-69): any other bit is refused by mifx_ssao_prepare */
typedef struct mifx_ssao_render_attribs /* ScreenSpaceAmbientOcclusion::RenderAttributes, .hpp:85-118 */
{
    mifx_postfx*             postfx;
    const mifx_image2d*      depth;  /* F32   */
    const mifx_image2d*      normal; /* F32X4, world space */
    const mifx_ssao_attribs* attribs;
} mifx_ssao_render_attribs;
MIFX_API mifx_status mifx_ssao_create(mifx_postfx* ctx, mifx_ssao** out);
MIFX_API void        mifx_ssao_destroy(mifx_ssao* fx);
MIFX_API mifx_status mifx_ssao_prepare(mifx_ssao* fx, mifx_postfx* ctx, uint32_t feature_flags); /* PrepareResources, .cpp:61 */
MIFX_API mifx_status mifx_ssao_execute(mifx_ssao* fx, const mifx_ssao_render_attribs* attribs);  /* Execute, .cpp:348 */
/* GetAmbientOcclusionSRV, .cpp:459: the resolved occlusion, ONE plane from mifx_ssao_prepare until a prepare that changes size / flags (a descriptor fetched once stays
 * valid across frames, like the reference's OCCLUSION_HISTORY_RESOLVED resource).  The effect object inside a mifx_chain is the exception: there the output aliases the
 * history plane of the last executed frame (it alternates with FrameDesc.Index & 1) and must be queried after every mifx_chain_execute. */
MIFX_API mifx_status mifx_ssao_get_output(mifx_ssao* fx, mifx_image2d* out);
MIFX_API mifx_status mifx_ssao_reset_history(mifx_ssao* fx);
/* Temporal state (no reference counterpart: the reference keeps it inside the object -- ping-pong by FrameDesc.Index & 1 and reset on a frame-index gap,
 * ScreenSpaceAmbientOcclusion.cpp:797-800, 1044-1045). export: copies the planes the NEXT frame will reproject (resolved AO, history length; F32, the prepared size) into
 * caller-owned device images on the context stream and returns the index of the frame that wrote them; MIFX_ERR_INVALID_OP when there is no history. import (after
 * mifx_ssao_prepare): overwrites them; the next execute with FrameDesc.Index == frame_index + 1 continues the accumulation as if this object had run frame_index. */
MIFX_API mifx_status mifx_ssao_export_history(mifx_ssao* fx, const mifx_image2d* out_ao, const mifx_image2d* out_history_length, uint32_t* out_frame_index);
MIFX_API mifx_status mifx_ssao_import_history(mifx_ssao* fx, const mifx_image2d* ao, const mifx_image2d* history_length, uint32_t frame_index);
/* Inspection of the effect-owned intermediates of the last execute (per-pass parity tests, debugging). Names:
 * "prefiltered_depth<1..4>", "occlusion", "history_ao", "history_len" (current slot), "conv_ao<1..4>", "conv_depth<1..4>", "resampled". */
MIFX_API mifx_status mifx_ssao_get_intermediate(mifx_ssao* fx, const char* name, mifx_image2d* out);
/* Test hook: A7 + A8 as one resolve (default: A7's copy and A8's early path inside the temporal pass A5, the pyramid walk and the spatial filter over work lists of the
 * texels that need them) or as the reference's two full-frame passes; every texel of the output and of every intermediate gets the same bits either way. */
MIFX_API mifx_status mifx_debug_ssao_set_fused_resolve(mifx_ssao* fx, int32_t enable);

/* ------------------------------------------------------------------------------------------------ ScreenSpaceReflection */
typedef struct mifx_ssr mifx_ssr; /* ScreenSpaceReflection.hpp:62-250 */
enum
{
    MIFX_SSR_FEATURE_FLAG_NONE            = 0,
    MIFX_SSR_FEATURE_FLAG_PREVIOUS_FRAME  = 1 << 0, /* `color` is last frame's: hits are reprojected with the motion vectors (SSR_ComputeIntersection.fx:310-314) */
    MIFX_SSR_FEATURE_FLAG_HALF_RESOLUTION = 1 << 1  /* half-size mask (R3) and ray pass (R4), R5 reads the half-size ray textures */
};
typedef struct mifx_ssr_render_attribs /* ScreenSpaceReflection::RenderAttributes, .hpp:89-121 */
{
    mifx_postfx*            postfx;
    const mifx_image2d*     color;    /* F32X4 scene radiance */
    const mifx_image2d*     depth;    /* F32   */
    const mifx_image2d*     normal;   /* F32X4 world space */
    const mifx_image2d*     material; /* F32X4, roughness in channel attribs->RoughnessChannel */
    const mifx_image2d*     motion;   /* F32X2 */
    const mifx_ssr_attribs* attribs;
} mifx_ssr_render_attribs;
MIFX_API mifx_status mifx_ssr_create(mifx_postfx* ctx, mifx_ssr** out);
MIFX_API void        mifx_ssr_destroy(mifx_ssr* fx);
MIFX_API mifx_status mifx_ssr_prepare(mifx_ssr* fx, mifx_postfx* ctx, uint32_t feature_flags); /* .cpp:67  */
MIFX_API mifx_status mifx_ssr_execute(mifx_ssr* fx, const mifx_ssr_render_attribs* attribs);   /* .cpp:300 */
MIFX_API mifx_status mifx_ssr_get_output(mifx_ssr* fx, mifx_image2d* out);                     /* GetSSRRadianceSRV, .cpp:460 */
/* Inside a mifx_chain with MIFX_CHAIN_FUSE_SSR_CLEANUP_INTO_COMPOSITE (the default) the effect stops after its temporal pass and the chain's composite evaluates the
 * bilateral cleanup per pixel: the output plane is then not written and mifx_ssr_get_output returns MIFX_ERR_INVALID_OP.  This call runs the deferred pass for the last
 * executed frame; `depth` / `normal` are that frame's planes again (the effect borrowed them for the execute only).  A no-op when nothing is deferred. */
MIFX_API mifx_status mifx_ssr_run_deferred_cleanup(mifx_ssr* fx, const mifx_image2d* depth, const mifx_image2d* normal);
MIFX_API mifx_status mifx_ssr_reset_history(mifx_ssr* fx);
/* Temporal state, as mifx_ssao_export_history / _import_history: accumulated radiance (F32X4) and variance (F32) of R6 (ping-pong ScreenSpaceReflection.cpp:1045-1046).
 * R5 and R6 run under the reflection mask and, like the reference's depth-tested draws, leave texels outside it as an earlier frame wrote them (.cpp:904-932, 1001-1069); R6 and R7
 * read such texels beside the mask's edge.  The COMPLETE temporal state is therefore both history slots and R5's three targets: export / import the frame before last as well (an
 * import with its index fills the other slot; import the newer frame last) and copy "res_radiance" / "res_variance" / "res_depth" (mifx_ssr_get_intermediate).  With the newest slot
 * alone the first frame after an import differs beside the mask's edge from a run that never moved (measured: 1.9e-3 of the SSR values of that one frame). */
MIFX_API mifx_status mifx_ssr_export_history(mifx_ssr* fx, const mifx_image2d* out_radiance, const mifx_image2d* out_variance, uint32_t* out_frame_index);
MIFX_API mifx_status mifx_ssr_import_history(mifx_ssr* fx, const mifx_image2d* radiance, const mifx_image2d* variance, uint32_t frame_index);
/* Names: "hiz<1..6>", "roughness", "mask", "ray_radiance", "ray_dir_pdf", "res_radiance", "res_variance", "res_depth",
 * "hist_radiance", "hist_variance" (current slot). */
MIFX_API mifx_status mifx_ssr_get_intermediate(mifx_ssr* fx, const char* name, mifx_image2d* out);

/* ------------------------------------------------------------------------------------------------ TemporalAntiAliasing */
typedef struct mifx_taa mifx_taa; /* TemporalAntiAliasing.hpp:60-214 */
enum
{
    MIFX_TAA_FEATURE_FLAG_NONE               = 0,
    MIFX_TAA_FEATURE_FLAG_GAUSSIAN_WEIGHTING = 1 << 0,
    MIFX_TAA_FEATURE_FLAG_BICUBIC_FILTER     = 1 << 1,
    MIFX_TAA_FEATURE_FLAG_YCOCG_COLOR_SPACE  = 1 << 2
};
typedef struct mifx_taa_render_attribs /* TemporalAntiAliasing::RenderAttributes, .hpp:84-110 */
{
    mifx_postfx*            postfx;
    const mifx_image2d*     color; /* F32X4 */
    const mifx_taa_attribs* attribs;
} mifx_taa_render_attribs;
MIFX_API mifx_status mifx_taa_create(mifx_postfx* ctx, mifx_taa** out);
MIFX_API void        mifx_taa_destroy(mifx_taa* fx);
MIFX_API mifx_status mifx_taa_prepare(mifx_taa* fx, mifx_postfx* ctx, uint32_t feature_flags); /* .cpp:145 */
MIFX_API mifx_status mifx_taa_execute(mifx_taa* fx, const mifx_taa_render_attribs* attribs);   /* .cpp:169 */
MIFX_API mifx_status mifx_taa_get_output(mifx_taa* fx, int32_t is_prev_frame, mifx_image2d* out); /* GetAccumulatedFrameSRV, .cpp:203 */
/* No reference counterpart: back to the state of a newly created object as far as results go. The next frame has no history and is, like the first frame of a new
 * TemporalAntiAliasing object in the reference, the placeholder frame of its flag set: the colour buffer copied into the accumulation buffer, alpha included
 * (the reference evaluates the readiness of a flag set's technique in PrepareResources and creates it in Execute: TemporalAntiAliasing.cpp:157-166, 187, 193-200, 290-300;
 * mifx_taa_execute returns MIFX_NO_HISTORY for it). A frame-index gap or ResetAccumulation resets through the shader instead, as in the reference. */
MIFX_API mifx_status mifx_taa_reset_history(mifx_taa* fx);
/* Temporal state, as mifx_ssao_export_history / _import_history: the accumulation buffer (F32X4, alpha = accumulated weight; TemporalAntiAliasing.cpp:123-143, 272-274). */
MIFX_API mifx_status mifx_taa_export_history(mifx_taa* fx, const mifx_image2d* out_color, uint32_t* out_frame_index);
MIFX_API mifx_status mifx_taa_import_history(mifx_taa* fx, const mifx_image2d* color, uint32_t frame_index);
/* Halton(2,3) jitter in NDC units, 16-sample cycle -- TemporalAntiAliasing::GetJitterOffset, .cpp:63-78 (static form) */
MIFX_API mifx_status mifx_taa_get_jitter_offset(uint32_t frame_index, uint32_t width, uint32_t height, float out_jitter[2]);
/* GetJitteredProjMatrix, TemporalAntiAliasing.hpp:138-155 */
MIFX_API mifx_status mifx_taa_get_jittered_proj_matrix(const float proj[16], const float jitter[2], float out_proj[16]);

/* ------------------------------------------------------------------------------------------------ Bloom */
typedef struct mifx_bloom mifx_bloom; /* Bloom.hpp:58-150 */
typedef struct mifx_bloom_render_attribs /* Bloom::RenderAttributes, Bloom.hpp:70-96 */
{
    mifx_postfx*              postfx;
    const mifx_image2d*       color; /* F32X4 */
    const mifx_bloom_attribs* attribs;
} mifx_bloom_render_attribs;
MIFX_API mifx_status mifx_bloom_create(mifx_postfx* ctx, mifx_bloom** out);
MIFX_API void        mifx_bloom_destroy(mifx_bloom* fx);
MIFX_API mifx_status mifx_bloom_prepare(mifx_bloom* fx, mifx_postfx* ctx, uint32_t feature_flags); /* Bloom.cpp:74  */
MIFX_API mifx_status mifx_bloom_execute(mifx_bloom* fx, const mifx_bloom_render_attribs* attribs); /* Bloom.cpp:407 */
MIFX_API mifx_status mifx_bloom_get_output(mifx_bloom* fx, mifx_image2d* out);                     /* GetBloomTextureSRV, Bloom.cpp:448 */
/* Names: "down<i>", "up<i>" (levels of the last execute). */
/* Test hook: the small pyramid levels in one workgroup (default) or one dispatch per level; the results are bit-identical. */
MIFX_API mifx_status mifx_debug_bloom_set_tail(mifx_bloom* fx, int32_t enable);
MIFX_API mifx_status mifx_bloom_get_intermediate(mifx_bloom* fx, const char* name, mifx_image2d* out);

/* ------------------------------------------------------------------------------------------------ DepthOfField */
typedef struct mifx_dof mifx_dof; /* PostProcess/DepthOfField/interface/DepthOfField.hpp:52-240 */
enum /* DepthOfField::FEATURE_FLAGS, DepthOfField.hpp:59-68 */
{
    MIFX_DOF_FEATURE_FLAG_NONE                      = 0u,
    MIFX_DOF_FEATURE_FLAG_ENABLE_TEMPORAL_SMOOTHING = 1u << 0,
    MIFX_DOF_FEATURE_FLAG_ENABLE_KARIS_INVERSE      = 1u << 1
};
typedef struct mifx_dof_render_attribs /* DepthOfField::RenderAttributes, DepthOfField.hpp:71-96 */
{
    mifx_postfx*            postfx;  /* camera (GetCameraAttribsCB), closest motion vectors, frame index */
    const mifx_image2d*     color;   /* F32X4 scene colour (pColorBufferSRV)  */
    const mifx_image2d*     depth;   /* F32 hardware depth (pDepthBufferSRV)  */
    const mifx_dof_attribs* attribs; /* pDOFAttribs                            */
} mifx_dof_render_attribs;
MIFX_API mifx_status mifx_dof_create(mifx_postfx* ctx, mifx_dof** out);                             /* DepthOfField.cpp:96-171: builds the Gauss and small Octaweb tables */
MIFX_API void        mifx_dof_destroy(mifx_dof* fx);
MIFX_API mifx_status mifx_dof_prepare(mifx_dof* fx, mifx_postfx* ctx, uint32_t feature_flags);      /* PrepareResources, DepthOfField.cpp:175-293 */
MIFX_API mifx_status mifx_dof_execute(mifx_dof* fx, const mifx_dof_render_attribs* attribs);        /* Execute, DepthOfField.cpp:295-332 */
MIFX_API mifx_status mifx_dof_get_output(mifx_dof* fx, mifx_image2d* out);                          /* GetDepthOfFieldTextureSRV: F32X4, a = alpha of the colour input (native-storage build: R11G11B10, which mifx_bloom_execute and the tone map accept) */
/* Planes after the last execute: "coc" (D1), "coc_temporal" (D2, current slot), "dilation1".."dilation3" (D4), "dilation_blurred" (D5),
 * "prefiltered0/1" (near / far: D6, overwritten by D8 as in the reference), "bokeh0/1" (D7, overwritten by D9). */
MIFX_API mifx_status mifx_dof_get_intermediate(mifx_dof* fx, const char* name, mifx_image2d* out);
/* No reference counterpart: DepthOfField has no reset -- its temporal circle of confusion blends with the previous slot on every frame and is only ever cleared when the targets
 * are (re)created (DepthOfField.cpp:205-223).  This clears both slots the same way, so that an effect with a past continues like a freshly prepared one
 * (mifx_chain_reset_history calls it for the chain's depth of field: "from a history reset" then means the same for every effect of the chain). */
MIFX_API mifx_status mifx_dof_reset_history(mifx_dof* fx);
/* Test hook: execute stops after pass `last_pass` (1 = D1 .. 10 = D10; 0 = run everything), so that the planes D8 / D9 overwrite can be read. */
MIFX_API mifx_status mifx_debug_dof_set_last_pass(mifx_dof* fx, uint32_t last_pass);
/* The Octaweb kernel the bokeh gather uses for (ring_count, ring_density) -- GenerateKernelPoints, DepthOfField.cpp:49-74; out: 2 * count floats,
 * count = 1 + density * (rings - 1) * rings / 2 <= 128 (the width of the reference's kernel texture). Host only. */
MIFX_API mifx_status mifx_dof_generate_kernel_points(int32_t ring_count, int32_t ring_density, float* out, uint32_t capacity_points, uint32_t* out_count);

/* ------------------------------------------------------------------------------------------------ PBR shading entry (lighting half of RenderPBR.psh:421-656) */
typedef struct mifx_gbuffer /* G-buffer contract: PBR/src/USD_Renderer.cpp:83-162, Hydrogent/src/Tasks/HnBeginFrameTask.cpp:63-69 */
{
    const mifx_image2d* base_color; /* F32X4 rgb = base colour, a = opacity                                  */
    const mifx_image2d* normal;     /* F32X4 xyz = world-space shading normal                                */
    const mifx_image2d* material;   /* F32X4 x = perceptual roughness, y = metallic (USD_Renderer.cpp:146-151) */
    const mifx_image2d* depth;      /* F32 hardware depth, non-reversed, background = 1                      */
    const mifx_image2d* emissive;   /* F32X4 or NULL                                                         */
    const mifx_image2d* occlusion;  /* F32 material AO or NULL (= 1)                                         */
} mifx_gbuffer;
typedef struct mifx_ibl /* PBR_Renderer::PrecomputeBRDF / PrecomputeCubemaps outputs, PBR_Renderer.cpp:548,729 */
{
    const mifx_image2d* brdf_lut;       /* F32X2 or F32X4 (rg used), PrecomputeBRDF.psh:41          */
    const mifx_cubemap* irradiance;     /* ComputeIrradianceMap.psh:85                               */
    const mifx_cubemap* prefiltered;    /* PrefilterEnvMap.psh:101, mip k <-> roughness k/(mips-1)   */
} mifx_ibl;
/* One invocation per pixel of GetSurfaceShadingInfo -> ApplyPunctualLight x N -> ApplyIBL -> ResolveLighting
 * (RenderPBR.psh:473-514, PBR_Shading.fxh:601-876). Background pixels (depth >= 1-1e-6) receive `background` colour.
 * out_specular_ibl receives GetBaseLayerSpecularIBL (the USD G-buffer "IBL" target, USD_Renderer.cpp:152-157); may be NULL. */
MIFX_API mifx_status mifx_pbr_shade_execute(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_camera_attribs* camera,
                                            const mifx_pbr_shade_attribs* attribs, const mifx_ibl* ibl, const float background[4],
                                            const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl);

/* The same shade with shadow-mapped punctual lights (ENABLE_SHADOWS of RenderPBR.psh:70-73,488-497): a light whose ShadowMapIndex is >= 0 is attenuated by
 * FilterShadowMapFixedPCF (Shaders/Common/public/PCF.fxh:7-152) of the shadow-map array at the light-space position of the pixel (PBR_Shading.fxh:644-660). */
#define MIFX_PBR_MAX_SHADOW_MAPS 8 /* PBR_Renderer::CreateInfo::MaxShadowCastingLightCount, PBR/interface/PBR_Renderer.hpp:248 */
typedef struct mifx_pbr_shadow_map_info /* PBRShadowMapInfo, Shaders/PBR/public/PBR_Structures.fxh:336-347 (96 bytes) */
{
    float WorldToLightProjSpace[16];
    float UVScale[2], UVBias[2];
    float ShadowMapSlice;
    float Padding0, Padding1, Padding2;
} mifx_pbr_shadow_map_info;
typedef struct mifx_shadow_map_array /* Texture2DArray<float> g_ShadowMap, sampled with Sam_ComparisonLinearClamp (comparison LESS) */
{
    const void* data;               /* F32 depth, slice k at data + k * slice_pitch_bytes */
    uint32_t    width, height, slices;
    uint32_t    pitch_bytes;
    uint64_t    slice_pitch_bytes;
} mifx_shadow_map_array;
typedef struct mifx_pbr_shadows
{
    const mifx_shadow_map_array*    shadow_map;
    const mifx_pbr_shadow_map_info* shadow_maps;      /* PBRFrameAttribs::ShadowMaps (RenderPBR_Structures.fxh:22), indexed by PBRLightAttribs::ShadowMapIndex */
    uint32_t                        shadow_map_count; /* <= MIFX_PBR_MAX_SHADOW_MAPS */
    uint32_t                        pcf_filter_size;  /* PCF_FILTER_SIZE: 2, 3 (PCFKernelSize default, PBR_Renderer.hpp:227), 5 or 7 */
} mifx_pbr_shadows;
MIFX_API mifx_status mifx_pbr_shade_execute_with_shadows(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_camera_attribs* camera,
                                                         const mifx_pbr_shade_attribs* attribs, const mifx_ibl* ibl, const mifx_pbr_shadows* shadows,
                                                         const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl);

/* The same shade with the material layers of the reference's PBR library (round 4): clear coat, sheen, anisotropy, iridescence, transmission -- the
 * ENABLE_CLEAR_COAT / ENABLE_SHEEN / ENABLE_ANISOTROPY / ENABLE_IRIDESCENCE / ENABLE_TRANSMISSION blocks of Shaders/PBR/public/PBR_Shading.fxh:40-62 (a pipeline
 * permutation per set of PSO flags in the reference, PBR/interface/PBR_Renderer.hpp:159-179, PBR_Renderer.cpp:1511-1516; all off by default).  `flags` is that set: a
 * layer that is off takes exactly the arithmetic of the permutation without it.  The planes carry what the fetches of GetSurfaceShadingInfo return per pixel
 * (RenderPBR.psh:186-297, PBR_Textures.fxh: texture x factor), as the G-buffer carries the base layer's.  "F32X4" below is the 4-channel texel of the build (F16X4 in
 * the native-storage build, like the G-buffer's planes); the one-channel plane and the tables are F32 in both builds. */
#define MIFX_PBR_LAYER_CLEAR_COAT   1u  /* KHR_materials_clearcoat:    ReadClearcoatLayerProperties RenderPBR.psh:186-220, ResolveLighting PBR_Shading.fxh:858-873 */
#define MIFX_PBR_LAYER_SHEEN        2u  /* KHR_materials_sheen:        ReadSheenLayerProperties :222-234, ApplyDirectionalLightSheen PBR_Shading.fxh:133, GetSpecularIBL_Charlie :347 */
#define MIFX_PBR_LAYER_ANISOTROPY   4u  /* KHR_materials_anisotropy:   ReadAnisotropyProperties :257-297, SmithGGX_BRDF_Anisotropic PBR_Common.fxh:407, bent normal PBR_Shading.fxh:754-767 */
#define MIFX_PBR_LAYER_IRIDESCENCE  8u  /* KHR_materials_iridescence:  ReadIridescenceProperties :236-255, Shaders/PBR/private/Iridescence.fxh */
#define MIFX_PBR_LAYER_TRANSMISSION 16u /* KHR_materials_transmission: the diffuse terms scaled by 1 - Transmission, PBR_Shading.fxh:694-698,748-752 */
typedef struct mifx_pbr_layers
{
    uint32_t            flags;               /* MIFX_PBR_LAYER_*; a plane is read only when its layer is on */
    float               iridescence_ior;     /* PBRMaterialIridescenceAttribs::IOR (Material.Iridescence.IOR, RenderPBR.psh:247)            */
    float               anisotropy_rotation; /* PBRMaterialAnisotropyAttribs::Rotation, radians (Material.Anisotropy.Rotation, :263)        */
    uint32_t            padding;
    const mifx_image2d* clearcoat;        /* F32X4: x = GetClearcoatFactor, y = GetClearcoatRoughness (the layer's IOR is 1.5, :198)                                        */
    const mifx_image2d* clearcoat_normal; /* F32X4 xyz = world-space normal of the layer (PerturbNormal's result, :208-215), or NULL = the G-buffer normal (no normal map)  */
    const mifx_image2d* sheen;            /* F32X4: rgb = GetSheenColor, a = GetSheenRoughness                                                                              */
    const mifx_image2d* anisotropy;       /* F32X4: xy = direction, z = strength (GetAnisotropy's packed value, before the material's rotation)                            */
    const mifx_image2d* tangent;          /* F32X4 xyz = world-space tangent (VSOut.Tangent, USE_VERTEX_TANGENTS), or NULL = (1, 0, 0) (:274-286)                         */
    const mifx_image2d* iridescence;      /* F32X4: x = GetIridescence (factor), y = GetIridescenceThickness in nm                                                         */
    const mifx_image2d* transmission;     /* F32: GetTransmission                                                                                                          */
    const mifx_image2d* sheen_albedo_scaling_lut; /* F32, F32X2 or F32X4 (r used): g_SheenAlbedoScalingLUT, loaded from a file by the reference (PBR_Renderer.cpp:407-423); sheen only */
    const mifx_image2d* preintegrated_charlie;    /* F32, F32X2 or F32X4 (r used): g_PreintegratedCharlie (PBR_Renderer.cpp:431-451); sheen only                                    */
} mifx_pbr_layers;
/* `shadows` may be NULL (no light has a ShadowMapIndex >= 0); otherwise as mifx_pbr_shade_execute_with_shadows: ENABLE_SHADOWS combines with every layer set. */
MIFX_API mifx_status mifx_pbr_shade_execute_layers(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const mifx_pbr_layers* layers, const mifx_camera_attribs* camera,
                                                   const mifx_pbr_shade_attribs* attribs, const mifx_ibl* ibl, const mifx_pbr_shadows* shadows, const float background[4],
                                                   const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl);

/* The reference's own constant blocks of the shade, byte for byte (round 3), and the entry that takes them as the renderer holds them.
 * PBRFrameAttribs (Shaders/PBR/private/RenderPBR_Structures.fxh:11-24) is a block whose tail depends on two compile-time limits of the renderer:
 *     CameraAttribs Camera (576 B) | CameraAttribs PrevCamera (576 B) | PBRRendererShaderParameters Renderer (144 B) |
 *     PBRLightAttribs Lights[PBR_MAX_LIGHTS] (64 B each, when PBR_MAX_LIGHTS > 0) | PBRShadowMapInfo ShadowMaps[PBR_MAX_SHADOW_MAPS] (96 B each, with ENABLE_SHADOWS)
 * -- the content of the cbPBRFrameAttribs / cbFrameAttribs buffer a PBR_Renderer user fills every frame (PBR_Renderer.hpp:245-248 for the two limits). */
typedef struct mifx_pbr_loading_animation_parameters /* LoadingAnimationShaderParameters, Shaders/PBR/public/PBR_Structures.fxh:111-120 (48 bytes) */
{
    float Factor, WorldScale, Speed, Padding;
    float Color0[4], Color1[4];
} mifx_pbr_loading_animation_parameters;
typedef struct mifx_pbr_renderer_shader_parameters /* PBRRendererShaderParameters, PBR_Structures.fxh:126-149 (144 bytes) */
{
    float   AverageLogLum, MiddleGray, WhitePoint;
    float   PrefilteredCubeLastMip;
    float   IBLScale[4];
    float   OcclusionStrength, EmissionScale;
    float   PointSize, MipBias;
    int32_t LightCount;
    float   Time;
    int32_t DebugView;
    float   Padding0;
    float   UnshadedColor[4], HighlightColor[4];
    mifx_pbr_loading_animation_parameters LoadingAnimation;
} mifx_pbr_renderer_shader_parameters;
typedef struct mifx_pbr_material_basic_attribs /* PBRMaterialBasicAttribs, PBR_Structures.fxh:154-180 (96 bytes) */
{
    float   BaseColorFactor[4];
    float   EmissiveFactorR, EmissiveFactorG, EmissiveFactorB, NormalScale;
    float   SpecularFactorR, SpecularFactorG, SpecularFactorB, ClearcoatNormalScale;
    int32_t Workflow, AlphaMode;
    float   AlphaMaskCutoff, MetallicFactor;
    float   RoughnessFactor, OcclusionFactor, ClearcoatFactor, ClearcoatRoughnessFactor;
    float   CustomData[4];
} mifx_pbr_material_basic_attribs;
/* What the entry below reads out of a PBRFrameAttribs block: Camera -> *out_camera; Renderer.{IBLScale, OcclusionStrength, EmissionScale, PrefilteredCubeLastMip,
 * LightCount} and Lights[0 .. LightCount) -> *out_attribs (the other Renderer fields drive passes outside this path: tone mapping inside the forward pass, debug
 * views, the loading animation; DebugView != 0 is refused); ShadowMaps -> out_shadow_maps[max_shadow_maps] when given; material->Workflow -> out_attribs->Workflow
 * (NULL material: metallic-roughness).  MIFX_ERR_INVALID_ARG when `frame_attribs_bytes` is not exactly the size of the block for these two limits. */
MIFX_API mifx_status mifx_pbr_shade_attribs_from_frame_attribs(const void* frame_attribs, uint64_t frame_attribs_bytes, uint32_t max_lights, uint32_t max_shadow_maps,
                                                               const mifx_pbr_material_basic_attribs* material, mifx_pbr_shade_attribs* out_attribs,
                                                               mifx_camera_attribs* out_camera, mifx_pbr_shadow_map_info* out_shadow_maps);
/* mifx_pbr_shade_execute / _with_shadows on the renderer's own frame block: `shadow_map` NULL = no shadows (ENABLE_SHADOWS off; max_shadow_maps must then be 0). */
MIFX_API mifx_status mifx_pbr_shade_execute_frame_attribs(mifx_postfx* ctx, const mifx_gbuffer* gbuffer, const void* frame_attribs, uint64_t frame_attribs_bytes, uint32_t max_lights,
                                                          uint32_t max_shadow_maps, const mifx_pbr_material_basic_attribs* material, const mifx_ibl* ibl,
                                                          const mifx_shadow_map_array* shadow_map, uint32_t pcf_filter_size, const float background[4],
                                                          const mifx_image2d* out_radiance, const mifx_image2d* out_specular_ibl);

/* The material constant block of a draw as the reference's renderer holds it -- PBRMaterialShaderInfo, Shaders/PBR/public/PBR_Structures.fxh:291-317:
 *     PBRMaterialBasicAttribs Basic (96 B) | Sheen (16 B, ENABLE_SHEEN) | Anisotropy (16 B, ENABLE_ANISOTROPY) | Iridescence (16 B, ENABLE_IRIDESCENCE) |
 *     Transmission (16 B, ENABLE_TRANSMISSION) | Volume (32 B, ENABLE_VOLUME) | PBRMaterialTextureAttribs Textures[PBR_NUM_TEXTURE_ATTRIBUTES] (48 B each)
 * -- the optional blocks are present when the pipeline was created with the layer's PSO flag.  Host only: reads the two per-material scalars of the layered shade out of it
 * (Iridescence.IOR -> iridescence_ior, Anisotropy.Rotation -> anisotropy_rotation), sets layers->flags = layer_flags, copies Basic to *out_basic when given (its Workflow
 * goes into mifx_pbr_shade_attribs::Workflow).  The planes of `layers` are not touched: they carry texture x factor per pixel, which the renderer's own fetches produce
 * (ClearcoatFactor / ClearcoatRoughnessFactor of Basic, the Sheen / Iridescence / Transmission factors: PBR_Textures.fxh).  `enable_volume`: the block carries
 * PBRMaterialVolumeAttribs as well (skipped: no lighting function reads it).  MIFX_ERR_INVALID_ARG when `bytes` is not exactly the size of the block for this set. */
typedef struct mifx_pbr_material_sheen_attribs /* PBRMaterialSheenAttribs, PBR_Structures.fxh:184-190 */ { float ColorFactorR, ColorFactorG, ColorFactorB, RoughnessFactor; } mifx_pbr_material_sheen_attribs;
typedef struct mifx_pbr_material_anisotropy_attribs /* PBRMaterialAnisotropyAttribs, :195-201 */ { float Strength, Rotation, Padding0, Padding1; } mifx_pbr_material_anisotropy_attribs;
typedef struct mifx_pbr_material_iridescence_attribs /* PBRMaterialIridescenceAttribs, :206-212 */ { float Factor, IOR, ThicknessMinimum, ThicknessMaximum; } mifx_pbr_material_iridescence_attribs;
typedef struct mifx_pbr_material_transmission_attribs /* PBRMaterialTransmissionAttribs, :217-223 */ { float Factor, Padding0, Padding1, Padding2; } mifx_pbr_material_transmission_attribs;
MIFX_API mifx_status mifx_pbr_layers_from_material_info(const void* material_info, uint64_t bytes, uint32_t layer_flags, int32_t enable_volume, uint32_t num_texture_attribs,
                                                        mifx_pbr_layers* layers, mifx_pbr_material_basic_attribs* out_basic);

/* IBL precompute == PBR_Renderer::PrecomputeBRDF (PBR_Renderer.cpp:548-622) and PBR_Renderer::PrecomputeCubemaps (:729-972).
 * The environment map is a float4 cube with a full (box-filtered) mip chain, as the reference expects of its input SRV. */
/* The shade samples the IBL cube maps through a working copy with a one-texel apron per face, made at every call because the maps are the caller's memory (they may
 * have been re-rendered since).  enable != 0: the caller declares the maps bound in mifx_ibl static -- the copy is then made once per set of maps (their addresses and
 * sizes) and again after any mifx_ibl_* call on this context, after this call, or when other maps are bound.  A caller that rewrites the same maps in place by other
 * means calls this function again.  Off by default. */
MIFX_API mifx_status mifx_postfx_set_static_ibl(mifx_postfx* ctx, int32_t enable);
MIFX_API mifx_status mifx_ibl_precompute_brdf_lut(mifx_postfx* ctx, const mifx_image2d* out_lut /* F32X2 */, uint32_t num_samples /* 512: PBR_Renderer.hpp:298 */);
/* One mip of the prefiltered environment map (PrefilterEnvMap.psh:40-98): out = out_size x 6*out_size float4 texels, tightly packed;
 * roughness = mip / (mip_count - 1) (PBR_Renderer.cpp:951); num_samples default 256 (:748-751). */
MIFX_API mifx_status mifx_ibl_prefilter_env_map(mifx_postfx* ctx, const mifx_cubemap* env, void* out, uint32_t out_size, float roughness, uint32_t num_samples);
/* Irradiance cube (ComputeIrradianceMap.psh:43-83): out = out_size x 6*out_size float4 texels; num_samples default 8192 on discrete GPUs (:627-664). */
MIFX_API mifx_status mifx_ibl_compute_irradiance_map(mifx_postfx* ctx, const mifx_cubemap* env, void* out, uint32_t out_size, uint32_t num_samples);
/* The same two passes from an equirectangular environment map (the ENV_MAP_TYPE_SPHERE permutation, PBR_Renderer.cpp:788-800): with roughness 0 the prefilter
 * pass is the reference's equirect -> cube conversion. */
MIFX_API mifx_status mifx_ibl_prefilter_env_map_sphere(mifx_postfx* ctx, const mifx_spheremap* env, void* out, uint32_t out_size, float roughness, uint32_t num_samples);
MIFX_API mifx_status mifx_ibl_compute_irradiance_map_sphere(mifx_postfx* ctx, const mifx_spheremap* env, void* out, uint32_t out_size, uint32_t num_samples);

/* Environment-map background == EnvMapRenderer::Prepare + Render (Components/src/EnvMapRenderer.cpp:204-277, interface/EnvMapRenderer.hpp:84-118):
 * every pixel at the far-plane depth receives the environment colour (Shaders/Common/private/EnvMap.psh:46-77; cube map, ENV_MAP_TYPE_CUBE) and, if
 * `motion` is given, its motion vector; other pixels are left untouched (depth test LESS_EQUAL, no depth writes). Hydrogent draws it with tone
 * mapping NONE, MipLevel 1, Alpha 0, motion vectors on (HnRenderEnvMapTask.cpp:165-219). */
enum /* EnvMapRenderer::OPTION_FLAGS */
{
    MIFX_ENVMAP_OPTION_FLAG_NONE                   = 0u,
    MIFX_ENVMAP_OPTION_FLAG_CONVERT_OUTPUT_TO_SRGB = 1u << 0, /* pow(colour, 1 / 2.2), EnvMap.psh:57-59 */
    MIFX_ENVMAP_OPTION_FLAG_COMPUTE_MOTION_VECTORS = 1u << 1,
    MIFX_ENVMAP_OPTION_FLAG_USE_REVERSE_DEPTH      = 1u << 2  /* depth test GREATER_EQUAL at the far plane (depth 0) */
};
typedef struct mifx_envmap_render_attribs /* EnvMapRenderer::RenderAttribs */
{
    const mifx_cubemap* env_map;         /* pEnvMap: float4 cube with its mip chain   */
    float               average_log_lum; /* AverageLogLum (1)                          */
    float               mip_level;       /* MipLevel (0)                               */
    float               alpha;           /* Alpha (1)                                  */
    uint32_t            options;         /* OPTION_FLAGS                               */
    float               scale[3];        /* Scale (1, 1, 1)                            */
    const mifx_spheremap* sphere_map;    /* used when env_map is NULL: ENV_MAP_TYPE_SPHERE (EnvMapRenderer.cpp:214-216) */
} mifx_envmap_render_attribs;
MIFX_API mifx_status mifx_envmap_render(mifx_postfx* ctx, const mifx_envmap_render_attribs* attribs, const mifx_tone_mapping_attribs* tone_mapping,
                                        const mifx_camera_attribs* camera, const mifx_camera_attribs* prev_camera, const mifx_image2d* depth,
                                        const mifx_image2d* color /* F32X4, read-modify-write */, const mifx_image2d* motion /* F32X2 or NULL */);

/* ------------------------------------------------------------------------------------------------ native formats at the boundary (first step)
 * Linear-layout images in the reference's texture formats (G-buffer: Hydrogent/src/Tasks/HnBeginFrameTask.cpp:63-69; effect outputs: R11G11B10_FLOAT,
 * R16_FLOAT, R16_UNORM; swap chain: RGBA8_UNORM_SRGB) <-> the fp32 planes of this library, with the conversion rules of a D3D11-class texture unit
 * / output merger (csrc/formats.hip). The effects themselves compute on fp32 planes. Enumerators are named after Diligent's TEX_FORMAT_*. */
enum
{
    MIFX_NATIVE_FORMAT_R32_FLOAT = 1, MIFX_NATIVE_FORMAT_RG32_FLOAT, MIFX_NATIVE_FORMAT_RGBA32_FLOAT,
    MIFX_NATIVE_FORMAT_R16_FLOAT, MIFX_NATIVE_FORMAT_RG16_FLOAT, MIFX_NATIVE_FORMAT_RGBA16_FLOAT,
    MIFX_NATIVE_FORMAT_R8_UNORM, MIFX_NATIVE_FORMAT_RG8_UNORM, MIFX_NATIVE_FORMAT_RGBA8_UNORM, MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB,
    MIFX_NATIVE_FORMAT_R16_UNORM, MIFX_NATIVE_FORMAT_RG16_UNORM, MIFX_NATIVE_FORMAT_RGBA16_UNORM,
    MIFX_NATIVE_FORMAT_R11G11B10_FLOAT
};
typedef struct mifx_native_image
{
    void*    data;        /* device memory, row-major, texels tightly packed within a row */
    uint32_t width, height, pitch_bytes;
    uint32_t format;      /* MIFX_NATIVE_FORMAT_* */
} mifx_native_image;
MIFX_API uint32_t    mifx_native_format_texel_size(uint32_t format); /* bytes; 0: unknown format */
/* dst: F32 / F32X2 / F32X4 plane of the same size; channels the source lacks read as (0, 0, 0, 1); a destination with fewer channels keeps the first ones */
MIFX_API mifx_status mifx_image_import(mifx_postfx* ctx, const mifx_native_image* src, const mifx_image2d* dst);
/* ToneMap() of the copy-frame pass (mifx_tonemap_execute) writing its render target's format directly -- what the output merger does for the reference when the
 * target is the swap chain (e.g. MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB; MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB is the shader-side conversion for non-sRGB targets,
 * HnCopyFrame.psh:61-63). Bit-identical to mifx_tonemap_execute followed by mifx_image_export. */
MIFX_API mifx_status mifx_tonemap_execute_native(mifx_postfx* ctx, const mifx_image2d* hdr_in, const mifx_native_image* ldr_out, const mifx_tone_mapping_attribs* attribs,
                                                 float ave_log_lum, uint32_t flags);
MIFX_API mifx_status mifx_image_export(mifx_postfx* ctx, const mifx_image2d* src, const mifx_native_image* dst);
/* mifx_pbr_shade_execute on the G-buffer as the reference stores it (HnBeginFrameTask.cpp:63-69: BaseColor RGBA8_UNORM, Normal RGBA16_FLOAT, Material RG8_UNORM,
 * depth D32_FLOAT read as R32_FLOAT) with the radiance / IBL targets in their own format (RGBA16_FLOAT there): the kernel decodes on load and encodes on store, no
 * fp32 copies of the planes exist. Any MIFX_NATIVE_FORMAT_* is accepted per plane (depth: R32_FLOAT). Bit-identical to mifx_image_import of every plane,
 * mifx_pbr_shade_execute, mifx_image_export of the outputs. Lights with shadow maps are not available on this entry. */
typedef struct mifx_gbuffer_native
{
    const mifx_native_image* base_color; /* rgb premultiplied base colour, a = opacity */
    const mifx_native_image* normal;     /* world-space shading normal in xyz          */
    const mifx_native_image* material;   /* x = perceptual roughness, y = metallic     */
    const mifx_native_image* depth;      /* R32_FLOAT                                  */
    const mifx_native_image* emissive;   /* or NULL                                    */
    const mifx_native_image* occlusion;  /* material AO in x, or NULL (= 1)            */
} mifx_gbuffer_native;
MIFX_API mifx_status mifx_pbr_shade_execute_native(mifx_postfx* ctx, const mifx_gbuffer_native* gbuffer, const mifx_camera_attribs* camera,
                                                   const mifx_pbr_shade_attribs* attribs, const mifx_ibl* ibl, const float background[4],
                                                   const mifx_native_image* out_radiance, const mifx_native_image* out_specular_ibl);

/* ------------------------------------------------------------------------------------------------ composite (Hydrogent/shaders/HnPostProcess.psh:145-185) */
typedef struct mifx_composite_attribs
{
    const mifx_image2d*              color;        /* F32X4 scene radiance (a = opacity)   */
    const mifx_image2d*              specular_ibl; /* F32X4                                */
    const mifx_image2d*              ssr;          /* F32X4 rgb radiance, a = confidence   */
    const mifx_image2d*              ssao;         /* F32                                  */
    const mifx_image2d*              normal;       /* F32X4                                */
    const mifx_image2d*              base_color;   /* F32X4                                */
    const mifx_image2d*              material;     /* F32X4 x = roughness, y = metallic    */
    const mifx_image2d*              brdf_lut;     /* preintegrated GGX                    */
    const mifx_camera_attribs*       camera;
    float                            ssr_scale;    /* PostProcessAttribs.SSRScale  (HnPostProcessStructures.fxh:4-21) */
    float                            ssao_scale;   /* PostProcessAttribs.SSAOScale */
    const mifx_tone_mapping_attribs* tone_mapping; /* NULL or mode NONE = no tone mapping (TAA on: HnPostProcessTask.cpp:172) */
    float                            ave_log_lum;
} mifx_composite_attribs;
/* The Material target a specular-glossiness surface contributes to the USD G-buffer: (PerceptualRoughness, Metallic) = (1 - glossiness, SolveMetallic(diffuse, specular)),
 * USD_Renderer.cpp:98 with GetSurfaceReflectance's specular-glossiness branch (PBR_Shading.fxh:93-117, 390-403). SSR and the composite read that plane; a caller with
 * specular-glossiness inputs produces it with this call. base_color / physical_desc / out_material: F32X4. */
MIFX_API mifx_status mifx_pbr_specgloss_to_material(mifx_postfx* ctx, const mifx_image2d* base_color, const mifx_image2d* physical_desc, const mifx_image2d* out_material);
MIFX_API mifx_status mifx_composite_execute(mifx_postfx* ctx, const mifx_composite_attribs* attribs, const mifx_image2d* out);

/* ------------------------------------------------------------------------------------------------ whole chain (the caller: HnPostProcessTask::Execute, Hydrogent/src/Tasks/HnPostProcessTask.cpp:743-948) */
typedef struct mifx_autoexposure mifx_autoexposure; /* auto exposure, declared below */
typedef struct mifx_chain mifx_chain;
typedef struct mifx_chain_frame
{
    mifx_frame_desc                  frame;
    mifx_gbuffer                     gbuffer;
    const mifx_image2d*              motion;      /* F32X2 */
    const mifx_image2d*              prev_depth;  /* F32   */
    const mifx_camera_attribs*       curr_camera;
    const mifx_camera_attribs*       prev_camera;
    const mifx_ibl*                  ibl;
    const mifx_pbr_shade_attribs*    pbr;
    const mifx_ssao_attribs*         ssao;
    const mifx_ssr_attribs*          ssr;
    const mifx_taa_attribs*          taa;
    const mifx_bloom_attribs*        bloom;
    const mifx_tone_mapping_attribs* tone_mapping;
    float                            ave_log_lum; /* 0.3: HnBeginFrameTask.cpp:648 */
    float                            ssr_scale, ssao_scale;
    float                            background[4];
    uint32_t                         taa_feature_flags;
    uint32_t                         tonemap_flags;
} mifx_chain_frame;
MIFX_API mifx_status mifx_chain_create(const mifx_device_desc* dev, const mifx_postfx_create_info* info, mifx_chain** out);
MIFX_API void        mifx_chain_destroy(mifx_chain* chain);
/* PBR shade -> prep -> SSR -> SSAO -> composite -> TAA -> Bloom -> ToneMap, recorded on the context stream. */
MIFX_API mifx_status mifx_chain_execute(mifx_chain* chain, const mifx_chain_frame* frame, const mifx_image2d* out_ldr);
MIFX_API mifx_status mifx_chain_get_postfx(mifx_chain* chain, mifx_postfx** out);
/* the effect objects the chain owns, by name: "ssao" (mifx_ssao*), "ssr" (mifx_ssr*), "taa" (mifx_taa*), "bloom" (mifx_bloom*), "dof" (mifx_dof*, NULL while off) -- for their outputs and intermediates */
MIFX_API mifx_status mifx_chain_get_effect(mifx_chain* chain, const char* name, void** out);
/* mifx_chain_execute with the final image in the copy-frame target's own format (e.g. MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB), see mifx_tonemap_execute_native.
 * Not available with a row band or together with mifx_chain_set_auto_exposure (MIFX_ERR_INVALID_ARG). */
MIFX_API mifx_status mifx_chain_execute_native(mifx_chain* chain, const mifx_chain_frame* f, const mifx_native_image* out_native);
MIFX_API mifx_status mifx_chain_reset_history(mifx_chain* chain); /* SSAO, SSR, TAA: their own reset rules; depth of field: its temporal circle of confusion cleared (mifx_dof_reset_history); auto exposure: the adapted average back at 0.1 (mifx_autoexposure_reset) */
/* Per-stage timing of the chain with HIP events recorded on the launch stream between the stages of mifx_chain_execute (the analogue of
 * the reference's ScopedDebugGroup markers, e.g. ScreenSpaceAmbientOcclusion.cpp:363). Stage order of `out_ms[MIFX_CHAIN_STAGE_COUNT]`:
 * pbr_shade, prep, ssr, ssao, composite, taa, dof (0 while off), bloom, tonemap. get_stage_times waits for the last executed frame. */
/* Auto exposure in the chain (off by default: Hydrogent passes a constant average): the final ToneMap takes fAveLogLum from the average
 * luminance of the Bloom output (mifx_autoexposure_*, elapsed time and adaptation as given here) instead of mifx_chain_frame::ave_log_lum. */
MIFX_API mifx_status mifx_chain_set_auto_exposure(mifx_chain* chain, int32_t enable, float elapsed_time_s, int32_t light_adaptation);
MIFX_API mifx_status mifx_chain_get_auto_exposure(mifx_chain* chain, mifx_autoexposure** out); /* NULL while off */
/* PostFXContext::FEATURE_FLAGS the chain prepares its context with (HnPostProcessTask.cpp:666-670: FEATURE_FLAG_REVERSED_DEPTH when the task
 * context says useReverseDepth); the shade's background test, SSR and SSAO follow it. */
MIFX_API mifx_status mifx_chain_set_postfx_feature_flags(mifx_chain* chain, uint32_t feature_flags);
/* FEATURE_FLAGS the chain prepares its SSAO / SSR objects with (HnPostProcessTaskParams::SSAOFeatureFlags / SSRFeatureFlags, HnPostProcessTask.cpp:672-678):
 * MIFX_SSAO_FEATURE_FLAG_HALF_RESOLUTION, MIFX_SSR_FEATURE_FLAG_PREVIOUS_FRAME (the chain hands SSR the shaded radiance of the current frame either way). */
MIFX_API mifx_status mifx_chain_set_effect_feature_flags(mifx_chain* chain, uint32_t ssao_feature_flags, uint32_t ssr_feature_flags);
/* Depth of field in the chain (off by default, like HnPostProcessTaskParams::EnableDOF): DepthOfField::Execute on the TAA output, Bloom then
 * reads its result (HnPostProcessTask.cpp:899-918). attribs == NULL turns it off. The effect object is mifx_chain_get_effect(chain, "dof"). */
MIFX_API mifx_status mifx_chain_set_depth_of_field(mifx_chain* chain, const mifx_dof_attribs* attribs, uint32_t feature_flags);
/* Material layers and shadow-mapped lights in the chain's shade (round 4): the frames executed from now on shade with mifx_pbr_shade_execute_layers(layers, shadows) instead of
 * mifx_pbr_shade_execute; either may be NULL, both NULL = the default shade again.  The structs, the image descriptors and the shadow-map infos they point to are copied; the
 * device planes are borrowed for every frame until the next call.  Lights with a ShadowMapIndex >= 0 need `shadows`.  SSR's pass R2 then runs as its own pass (the default
 * shade kernel writes its two planes as a by-product, this one does not).  With a row band the sharded SSR's hit fetch shades its pixels with the same layers and shadow maps
 * (every rank holds the planes whole, as it holds the G-buffer). */
MIFX_API mifx_status mifx_chain_set_material_layers(mifx_chain* chain, const mifx_pbr_layers* layers, const mifx_pbr_shadows* shadows);
/* Row-band sharding of one frame across the GPUs of a node. A chain with a row band [row_begin, row_end) produces those
 * rows of the output; every pass runs on the rows its consumers need (the band grown by the reach of everything downstream), the caller
 * moves two kinds of data (three with auto exposure) between the phases of mifx_chain_execute_phase (diligentfx_amd/tiling.py does it with RCCL):
 *   phases 0 (shade) and 1 (prep, SSAO) need nothing from the other ranks. (Until round 3 the band rows of "radiance" were all-gathered after phase 0 for the SSR
 *                  ray march; the march now records where every ray hit and phase 2 loads the colour there from the rows this rank shaded or shades that one
 *                  pixel itself -- the G-buffer and the IBL maps are whole on every rank -- bit-identical, and 465 MB per GPU and frame less at 8K / 8 ranks.)
 *   after phase 2 (SSR, composite, TAA, Bloom fine levels): "bloom_gather" (Bloom level `gather_level`): rows [own_begin, own_end) are
 *                  valid on this rank, every rank needs all rows;
 *   after phase 3 (Bloom, tone map): halo exchange of the history planes: the halo_* rows above and below this rank's band (its ghost rows) are replaced by
 *                  the rows of the ranks that own them -- the neighbours, and the ranks beyond them where a halo is taller than a neighbour's band.
 *   auto exposure on (mifx_chain_set_auto_exposure): phase 3 ends with the rows [ae_begin, ae_end) of the 64 x 64 low-resolution luminance "ae_low_res"
 *                  instead of the tone map; every rank needs all 64 rows, then phase 4 reduces them (the same values in the same order on every rank: one
 *                  average, bit-identical to the unsharded frame's) and tone-maps the band. Phase 4 does nothing without auto exposure.
 * `max_motion_rows` bounds the reprojection reach (|motion| in rows); row_begin = row_end = 0 switches sharding off. */
typedef struct mifx_shard_info {
    int32_t band_begin, band_end;
    int32_t halo_taa, halo_ssr, halo_ssao; /* rows of "taa_history" / "ssr_history_*" / "ssao_history_*" needed above and below the band */
    int32_t gather_level, own_begin, own_end;
    int32_t ae_begin, ae_end; /* auto exposure on: rows of "ae_low_res" (64 x 64) this rank writes in phase 3; both 0 otherwise */
} mifx_shard_info;
MIFX_API mifx_status mifx_chain_set_row_band(mifx_chain* chain, int32_t row_begin, int32_t row_end, int32_t max_motion_rows);
MIFX_API mifx_status mifx_chain_execute_phase(mifx_chain* chain, const mifx_chain_frame* frame, const mifx_image2d* out_ldr, int32_t phase);
MIFX_API mifx_status mifx_chain_get_shard_info(mifx_chain* chain, const mifx_chain_frame* frame, mifx_shard_info* out);
/* name: "radiance", "bloom_gather", "taa_history", "ssr_history_radiance", "ssr_history_variance", "ssao_history_ao", "ssao_history_len", "ae_low_res" */
MIFX_API mifx_status mifx_chain_get_shard_plane(mifx_chain* chain, const char* name, mifx_image2d* out);
/* The same frame with the exchanges done by the library (no reference counterpart: DiligentFX renders on one GPU). One process -- or one host thread -- per GPU:
 *   rank 0:      mifx_comm_get_unique_id(id); hand `id` to the other ranks by any side channel (file, environment, MPI, torch.distributed)
 *   every rank:  mifx_comm_create(ctx, id, rank, world, &comm)            = ncclCommInitRank on the context's device (collective call)
 *                mifx_chain_set_sharding(chain, comm, row_cuts, max_motion_rows)   row_cuts[world + 1]: 0 = cuts[0] < ... < cuts[world] = frame height
 *                every frame: mifx_chain_execute_sharded(chain, &frame, &out)      rows [cuts[rank], cuts[rank + 1]) of `out` are this rank's share of the image
 * mifx_chain_execute_sharded runs the phases above and moves those rows with grouped ncclSend / ncclRecv over xGMI (direct transfers between
 * the two ranks concerned). RCCL is loaded at the first mifx_comm call (dlopen); a missing library and
 * every RCCL failure are MIFX_ERR_COMM with the RCCL message in mifx_last_error(). world == 1 degenerates to mifx_chain_execute.
 * mifx_comm_create_local_group: `world` endpoints inside one process that share the context's device -- the same code path with device copies in place of RCCL,
 * each endpoint driven by its own host thread; for tests on a single GPU (RCCL refuses two ranks on one device). */
typedef struct mifx_comm mifx_comm;
#define MIFX_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */
MIFX_API mifx_status mifx_comm_get_unique_id(uint8_t out_id[MIFX_COMM_ID_BYTES]);
MIFX_API mifx_status mifx_comm_create(mifx_postfx* ctx, const uint8_t id[MIFX_COMM_ID_BYTES], int32_t rank, int32_t world, mifx_comm** out);
MIFX_API mifx_status mifx_comm_create_local_group(mifx_postfx* ctx, int32_t world, mifx_comm** out_comms /* [world] */);
MIFX_API void        mifx_comm_destroy(mifx_comm* comm);
MIFX_API mifx_status mifx_comm_get_info(const mifx_comm* comm, int32_t* out_rank, int32_t* out_world, int32_t* out_is_rccl);
/* First contact with the transport before a frame depends on it (collective: every rank calls it): every rank sends every peer `bytes_per_peer` bytes (a multiple of 4)
 * that name sender, receiver and word, through the same grouped ncclSend / ncclRecv the frames use, and verifies what it receives.  The host waits at most `timeout_ms`
 * for the exchange; a peer that never answers ends in ncclCommAbort and MIFX_ERR_COMM instead of a hang.  On failure mifx_last_error() carries RCCL's message; the caller
 * falls back to exchanges of its own (bench.py: --comm torch) or gives up. */
MIFX_API mifx_status mifx_comm_self_test(mifx_comm* comm, mifx_postfx* ctx, uint32_t bytes_per_peer, uint32_t timeout_ms);
/* What an endpoint is and what it has moved (round 6; no reference counterpart).  `ranks_in_communicator` is the transport's own answer -- ncclCommCount() of the RCCL
 * communicator, the size of an in-process group, -1 when the transport does not export the call -- so that a caller can tell "RCCL spans the N ranks I started" from `world`
 * echoed back.  bytes_* and groups count from creation (the self test's included).  mifx_comm_set_timing(comm, 1) brackets every exchange group of the frames that follow
 * with two timing events on the stream the group is issued on (start: the stream reaches the group; stop: its transfers are done on this rank, waiting for a late peer
 * included); mifx_comm_get_stats waits for those groups, reports their number, the sum and the largest of their durations, and forgets them.  Off by default. */
typedef struct mifx_comm_stats {
    int32_t  rank, world, is_rccl, ranks_in_communicator;
    uint64_t groups, bytes_sent, bytes_received;
    uint32_t timed_groups;
    float    exchange_ms_total, exchange_ms_max;
} mifx_comm_stats;
MIFX_API mifx_status mifx_comm_set_timing(mifx_comm* comm, int32_t enable);
MIFX_API mifx_status mifx_comm_get_stats(mifx_comm* comm, mifx_comm_stats* out);
MIFX_API mifx_status mifx_chain_set_sharding(mifx_chain* chain, mifx_comm* comm /* borrowed; NULL: off */, const int32_t* row_cuts, int32_t max_motion_rows);
MIFX_API mifx_status mifx_chain_execute_sharded(mifx_chain* chain, const mifx_chain_frame* frame, const mifx_image2d* out_ldr);
/* With mifx_chain_set_overlap >= 2 (and its input contract) a sharded frame runs as two lanes across frames: phases 0 - 2 and their exchanges on a side stream, phase 3 --
 * Bloom's coarse levels and the final pass, small launches that leave the GPU idle -- on the context's stream, beside the next frame's shade and SSAO (round 5).
 * With mifx_chain_set_overlap >= 3, two more: PostFX prep + SSAO (phase 1) on a lane of their own beside the shade and SSR (SSR waits for the prep, the composite for SSAO), and
 * SSR's depth hierarchy -- whole-frame work on every rank -- on a fourth stream beside the shade (the march waits for it).
 * mifx_chain_execute_band: the same phases and lanes for the band of mifx_chain_set_row_band WITHOUT the exchanges (ghost rows stale: the work is the same, the frame is not
 * an image) -- the compute side of one rank, for cost models and tools (tools/shard_cost.py). */
MIFX_API mifx_status mifx_chain_execute_band(mifx_chain* chain, const mifx_chain_frame* frame, const mifx_image2d* out_ldr);
/* PostFX prep + SSAO are independent of PBR shade + SSR until the composite. mifx_chain_set_overlap (or MIFX_CHAIN_OVERLAP=1|2|3|4|5 in the environment):
 *   1  the chain records them on a second stream and joins before the composite;
 *   2  and across frames: the second stream of the next frame waits only for this frame's last reader of what prep and SSAO overwrite (SSR, TAA, depth of
 *      field), not for its Bloom and tone map -- the next frame's prep + SSAO then fill the GPU under the small launches of the Bloom pyramid. The caller
 *      guarantees that a frame's input planes (G-buffer, depth, motion) are complete when mifx_chain_execute is called: they are read on a stream that does not
 *      wait for work queued earlier on the context's stream;
 *   3  three lanes by resource class, sliding against each other across frames (same input contract as 2):
 *        S  PBR shade, PostFX prep, SSR's depth hierarchy, SSAO          (vector-ALU bound kernels and their pyramids)
 *        X  SSR ray march / resolve / accumulation, composite, TAA, DOF  (latency- and bandwidth-bound kernels)
 *        M  Bloom + tone map on the context's stream                     (many small dependent launches)
 *      each lane waits only for what it reads: the ray march of a frame runs beside its ambient-occlusion pass, the Bloom pyramid beside the next frame's shade.
 *      When mifx_chain_execute returns, the context's stream is ordered behind all three lanes of the frame.
 *   4  the lanes of 3 with two frames in flight (round 5; same input contract): lane S of frame N + 1 starts when lane X of frame N - 1 has ended and so runs beside
 *      lane X of frame N -- the resolve, accumulation, composite and TAA kernels that run alone in mode 3. The planes S writes and X reads (radiance, specular IBL, SSR's
 *      roughness / mask / depth hierarchy, the PostFX planes, the blue noise) exist twice inside the chain and alternate from frame to frame (+57 B/px of memory);
 *      intermediates handed out after a frame (mifx_ssr_get_intermediate, mifx_postfx_get_*) are that frame's. A frame whose FrameDesc.Index does not follow its
 *      predecessor's is ordered like mode 3. mifx_chain_set_lane_edges (MIFX_LANE_EDGES) adds ordering between kernels of different lanes / frames in this mode.
 *   5  mode 4 with the frame's bandwidth-bound tail -- the composite (+ R7), TAA, depth of field -- on lane M in front of Bloom instead of at the end of lane X (round 6):
 *      lane X of a frame is the ray march, the resolve and the accumulation only, and the NEXT frame's march runs beside this frame's composite and TAA, which mode 4 left
 *      with little beside them. Same planes, same contract as 4 (with R7 as a pass of its own, fusion bit 2 off, the next frame's SSR waits for this frame's composite).
 *      At 4K on an MI355X: 1.0 - 1.2 % faster than mode 4; what bench.py runs for N = 1.
 * Work the library itself queues on the context's stream between two frames (mifx_chain_reset_history, mifx_*_import_history, a prepare that re-allocates, depth of
 * field switched on) is detected and ordered in front of every lane of the next frame.
 * Same kernels and bit-identical results in every mode; measured at 4K on an MI355X: 1.81 / 1.76 / 1.71 ms per frame (mode 0 / 1 / 2; mode 3: DESIGN.md section 4).
 * Off by default so that kernel durations stay attributable (two kernels sharing the GPU both look slower) and because of the contract of modes 2 and 3; ignored
 * while stage profiling is on. */
MIFX_API mifx_status mifx_chain_set_overlap(mifx_chain* chain, int32_t enable);
/* Modes 4 and 5 only; no reference counterpart. `edges` = "waiter<signal@frames,...": the kernel `waiter` of frame N does not start before the kernel `signal` of frame
 * N - frames (0 .. 3) is done; names are those of mifx_postfx_set_kernel_timing ("ssao_compute_ao_kernel<ssr_intersection_kernel@1": a frame's ambient-occlusion pass
 * starts when the previous frame's ray march is done, i.e. runs beside that frame's resolve / accumulation passes instead of beside its march). Ordering only: the
 * results do not depend on it; an edge whose signal that frame did not launch is ignored. NULL or "" removes all edges. */
MIFX_API mifx_status mifx_chain_set_lane_edges(mifx_chain* chain, const char* edges);
/* Pass fusion inside the chain (both on by default; the results are bit-identical either way -- the switches exist for A/B measurement and for the tests that say so):
 *   tone_map_into_bloom: the copy-frame ToneMap() is the tail of Bloom's final up-sample kernel (one read of the frame less; the "tonemap" stage time moves into "bloom");
 *                        applies to a plain fp32 target with a constant average luminance (not mifx_chain_execute_native / auto exposure);
 *   ssr_mask_into_shade: the shade kernel also writes ScreenSpaceReflection's roughness / reflection-mask planes (pass R2 reads the same material and depth texels);
 *                        with a row band the shade covers the few extra rows of the ray march. */
MIFX_API mifx_status mifx_chain_set_fusion(mifx_chain* chain, int32_t tone_map_into_bloom, int32_t ssr_mask_into_shade);
/* All fusion switches as one mask (MIFX_CHAIN_FUSE_DEFAULT is what a new chain has; mifx_chain_set_fusion sets the first two and leaves the others alone).  Every switch gives the same bits
 * of the final frame and of every history plane either way (tests/test_gpu_chain.py: test_chain_fusion_is_bit_identical):
 *   SSR_CLEANUP_INTO_COMPOSITE: ScreenSpaceReflection's last pass (R7, the bilateral cleanup) is evaluated per pixel inside the composite kernel, the only consumer of its
 *                        target; the effect's output plane is then produced on demand (mifx_ssr_run_deferred_cleanup) instead of every frame;
 *   SSAO_RESOLVE:        ScreenSpaceAmbientOcclusion's passes A7 + A8 folded into its temporal pass + two work-list passes (== mifx_debug_ssao_set_fused_resolve on the chain's effect object);
 *   BLOOM_OUTPUT_ON_DEMAND: with TONE_MAP_INTO_BLOOM in effect the final up-sample writes the tone-mapped frame only -- nothing in the chain reads the Bloom output plane
 *                        after it (16 B per pixel less) -- and mifx_bloom_get_output of the chain's Bloom object produces the plane when somebody asks for it, from the
 *                        frame just executed (valid until the next execute, like every effect output). */
enum
{
    MIFX_CHAIN_FUSE_TONE_MAP_INTO_BLOOM        = 1u << 0,
    MIFX_CHAIN_FUSE_SSR_MASK_INTO_SHADE        = 1u << 1,
    MIFX_CHAIN_FUSE_SSR_CLEANUP_INTO_COMPOSITE = 1u << 2,
    MIFX_CHAIN_FUSE_SSAO_RESOLVE               = 1u << 3,
    MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND     = 1u << 4,
    MIFX_CHAIN_FUSE_COMPOSITE_INTO_TAA         = 1u << 5, /* round 5, OFF by default (the one switch that is): the composite (with bit 2: SSR's cleanup inside) evaluated by the
                                                             TAA kernel for the texels of its colour tile instead of a pass of its own -- the composite plane is neither
                                                             written nor read; needs bit 2, and TAA's placeholder frame (the first of a flag set) still runs the composite
                                                             pass, whose plane it copies.  Bit-identical like the others, but measured slower on the MI355X (one kernel
                                                             420 us against 165 + 170 us at 3840x2160: the 1.33x re-evaluated halo and one pass' latency chain appended to
                                                             the other's at five instead of seven resident waves): kept as a measured alternative, DESIGN.md section 4 */
    MIFX_CHAIN_FUSE_DEFAULT                    = 0x1Fu,
    MIFX_CHAIN_FUSE_ALL                        = 0x1Fu, /* every fusion that pays = the default set (rounds 2 - 4's meaning; round 5 had widened it to the measured-slower bit 5) */
    MIFX_CHAIN_FUSE_EXPERIMENTAL               = 0x20u, /* the switches that are bit-identical but measured slower: off unless asked for by name */
    MIFX_CHAIN_FUSE_EVERY_SWITCH               = 0x3Fu  /* what mifx_chain_set_fusion_mask accepts (tests: every switch at once) */
};
MIFX_API mifx_status mifx_chain_set_fusion_mask(mifx_chain* chain, uint32_t mask);
#define MIFX_CHAIN_STAGE_COUNT 9
MIFX_API mifx_status mifx_chain_set_profiling(mifx_chain* chain, int32_t enable);
MIFX_API mifx_status mifx_chain_get_stage_times(mifx_chain* chain, float out_ms[MIFX_CHAIN_STAGE_COUNT]);

/* ------------------------------------------------------------------------------------------------ auto exposure
 * The average scene luminance that ToneMap() takes as fAveLogLum, computed the way the reference's light-scattering post-process does
 * (EpipolarLightScattering.cpp:2496-2506): a 64x64 low-resolution image of weighted log-luminance (GetWeightedLogLum,
 * AtmosphereShadersCommon.fxh:197-203, MinLuminance 0.01, of a linear-clamp sample of the scene colour at the texel centre --
 * UnwarpEpipolarScattering.fx:283-307 without in-scattering / extinction), its mip chain down to 1x1 (GenerateMips: 2x2 box), and
 * UpdateAverageLuminancePS (UpdateAverageLuminance.fx:12-29) alpha-blended into the 1x1 average (initially 0.1, .cpp:892-905).
 * One 1024-thread workgroup: the 2x2 box levels run on wave shuffles.  LOW_RES_LUMINANCE_MIPS = 7 (AtmosphereShadersCommon.fxh:54-56). */
MIFX_API mifx_status mifx_autoexposure_create(mifx_postfx* ctx, mifx_autoexposure** out);
MIFX_API void        mifx_autoexposure_destroy(mifx_autoexposure* ae);
/* light_adaptation != 0: the new value is weighted by 1 - exp(-elapsed_time_s) (LIGHT_ADAPTATION, fAdaptationRate = 1), else by 1 */
MIFX_API mifx_status mifx_autoexposure_execute(mifx_autoexposure* ae, const mifx_image2d* scene_color, float elapsed_time_s, int32_t light_adaptation);
MIFX_API mifx_status mifx_autoexposure_reset(mifx_autoexposure* ae, float average_luminance); /* the reference starts at 0.1 */
/* "average_luminance" (1x1 F32, g_tex2DAverageLuminance) or "low_res_luminance" (64x64 F32X2, mip 0 of g_tex2DLowResLuminance) */
MIFX_API mifx_status mifx_autoexposure_get_plane(mifx_autoexposure* ae, const char* name, mifx_image2d* out);
/* GetAverageSceneLuminance (AtmosphereShadersCommon.fxh:188-195): max(0.05, average); waits for the stream */
MIFX_API mifx_status mifx_autoexposure_get_average(mifx_autoexposure* ae, float* out);
/* ToneMap with fAveLogLum = GetAverageSceneLuminance() read on the device (no host round trip) */
MIFX_API mifx_status mifx_tonemap_execute_auto(mifx_postfx* ctx, const mifx_image2d* hdr_in, const mifx_image2d* ldr_out, const mifx_tone_mapping_attribs* attribs,
                                               mifx_autoexposure* ae, uint32_t flags);

/* ------------------------------------------------------------------------------------------------ misc */
MIFX_API uint32_t    mifx_abi_version(void);
MIFX_API uint32_t    mifx_sizeof(const char* struct_name); /* layout check for bindings: "camera_attribs", "ssao_attribs", ... */
/* Kernel timing with HIP events on the launch stream (the roofline figure of bench.py): arms a bracket around every later launch of
 * `kernel_name` -- one of "ssr_intersection_kernel", "ssr_spatial_kernel", "ssr_temporal_kernel",
 * "ssao_compute_ao_kernel", "taa_kernel", "composite_kernel", "bloom_upsample_kernel" (final pass), "tonemap_kernel" -- for up to `slots`
 * launches; NULL or 0 disarms. get_kernel_times waits for the recorded launches and returns their durations in launch order. */
MIFX_API mifx_status mifx_postfx_set_kernel_timing(mifx_postfx* ctx, const char* kernel_name, uint32_t slots);
MIFX_API mifx_status mifx_postfx_get_kernel_times(mifx_postfx* ctx, float* out_ms, uint32_t capacity, uint32_t* out_count);
/* Diagnostics (no reference counterpart): evaluates one device math helper of the kernels' fp32 policy element-wise on device arrays,
 * out[i] = op(a[i], b[i]), so that tests can hold the helpers to their stated accuracy (division and square root against IEEE, the
 * bounded sin / cos, the hardware exp / pow). `b` may be null for unary operations. */
typedef enum mifx_math_op {
    MIFX_MATH_FDIV = 0, MIFX_MATH_FSQRT = 1, MIFX_MATH_SIN_BOUNDED = 2, MIFX_MATH_COS_BOUNDED = 3, MIFX_MATH_EXP = 4, MIFX_MATH_POW = 5
} mifx_math_op;
/* Diagnostics: a device-to-device copy with the library's own streaming access pattern (one 16-byte texel per lane) on the context stream -- the achievable HBM
 * rate bench.py reports beside the spec peak.  src / dst / bytes: multiples of 16. */
MIFX_API mifx_status mifx_debug_stream_copy(mifx_postfx* ctx, const void* src, void* dst, uint64_t bytes);
MIFX_API mifx_status mifx_debug_eval_math(mifx_postfx* ctx, uint32_t op, const float* a, const float* b, float* out, uint64_t n);

#ifdef __cplusplus
} /* extern "C" */
#endif
#endif /* MIFX_H */
