// adapter_smoke.cpp -- drives the adapters through one frame's call protocol (PostFXContext::PrepareResources -> effects' PrepareResources ->
// PostFXContext::Execute -> effects' Execute, HnPostProcessTask.cpp:671-682,808-917) with stand-in interop.  It has no HIP dependency of its
// own: on a machine without a GPU mifx_postfx_create fails, every later call must degrade to a logged no-op (the reference's "log and return"
// error model) and the program must still exit 0 -- that is what tests/test_adapter_example.py checks on the CPU.
#include "mifx_effect_adapters.hpp"

#include <cstdio>
#include <cstring>

namespace Diligent
{
struct ITextureView // stand-in: a view is just the image descriptor it wraps
{
    mifx_image2d img;
};
namespace NoiseBuffers
{
const unsigned char Sobol_256d[256]                = {};
const unsigned char ScramblingTile[128 * 128 * 8] = {};
} // namespace NoiseBuffers

mifx_image2d GetMifxImage(ITextureView* pView, uint32_t Format)
{
    mifx_image2d img{};
    if (pView != nullptr) img = pView->img;
    img.format = Format;
    return img;
}
ITextureView* WrapMifxImage(const mifx_image2d& Image)
{
    static ITextureView views[16];
    static unsigned     next = 0;
    ITextureView&       v    = views[next++ % 16u];
    v.img                    = Image;
    return &v;
}
void* GetMifxStream(IDeviceContext*) { return nullptr; }
bool  GetMifxCubemap(ITextureView*, mifx_cubemap&) { return false; } // (no cube maps in this driver)
} // namespace Diligent

int main()
{
    using namespace Diligent;
    std::printf("mifx ABI %u; camera block %u bytes\n", mifx_abi_version(), mifx_sizeof("camera_attribs"));
    PostFXContext               postfx(nullptr, PostFXContext::CreateInfo{});
    ScreenSpaceAmbientOcclusion ssao(nullptr, {});
    ScreenSpaceReflection       ssr(nullptr, {});
    TemporalAntiAliasing        taa(nullptr, {});
    Bloom                       bloom(nullptr, {});
    DepthOfField                dof(nullptr, {});
    const bool haveDevice = postfx.GetMifxContext() != nullptr;

    PostFXContext::FrameDesc frame;
    frame.Index = 0;
    frame.Width = frame.OutputWidth = 64;
    frame.Height = frame.OutputHeight = 32;
    postfx.PrepareResources(nullptr, frame, PostFXContext::FEATURE_FLAG_NONE);
    ssao.PrepareResources(nullptr, nullptr, &postfx, ScreenSpaceAmbientOcclusion::FEATURE_FLAG_NONE);
    ssr.PrepareResources(nullptr, nullptr, &postfx, ScreenSpaceReflection::FEATURE_FLAG_NONE);
    taa.PrepareResources(nullptr, nullptr, &postfx, TemporalAntiAliasing::FEATURE_FLAG_BICUBIC_FILTER);
    bloom.PrepareResources(nullptr, nullptr, &postfx, Bloom::FEATURE_FLAG_NONE);
    dof.PrepareResources(nullptr, nullptr, &postfx, DepthOfField::FEATURE_FLAG_NONE);

    const float2 jitter = taa.GetJitterOffset(); // host arithmetic: works with or without a device
    std::printf("jitter of frame 0 at 64x32: (%g, %g)\n", jitter.x, jitter.y);

    HLSL::CameraAttribs cam{};
    PostFXContext::RenderAttributes pra;
    pra.pCurrCamera = &cam;
    pra.pPrevCamera = &cam;
    postfx.Execute(pra); // null views: rejected by the library, logged by the adapter

    HLSL::ScreenSpaceAmbientOcclusionAttribs ssaoAttribs{};
    ScreenSpaceAmbientOcclusion::RenderAttributes sra;
    sra.pPostFXContext = &postfx;
    sra.pSSAOAttribs   = &ssaoAttribs;
    ssao.Execute(sra);

    // the PBR shade of a G-buffer on the renderer's own frame block: the host half (block layout, light list) works without a device
    {
        struct FrameBlock // PBRFrameAttribs with PBR_MAX_LIGHTS = 2, ENABLE_SHADOWS off (RenderPBR_Structures.fxh:11-24)
        {
            HLSL::CameraAttribs                 Camera, PrevCamera;
            mifx_pbr_renderer_shader_parameters Renderer;
            mifx_pbr_light_attribs              Lights[2];
        } block{};
        block.Renderer.IBLScale[0] = block.Renderer.IBLScale[1] = block.Renderer.IBLScale[2] = 1.0f;
        block.Renderer.PrefilteredCubeLastMip = 8.0f;
        block.Renderer.LightCount             = 1;
        block.Lights[0].Type                  = MIFX_PBR_LIGHT_TYPE_DIRECTIONAL;
        block.Lights[0].ShadowMapIndex        = -1;
        mifx_pbr_shade_attribs attribs{};
        mifx_camera_attribs    camera{};
        const mifx_status st = mifx_pbr_shade_attribs_from_frame_attribs(&block, sizeof(block), 2, 0, nullptr, &attribs, &camera, nullptr);
        std::printf("PBRFrameAttribs (%zu bytes, 2 lights): %s, LightCount %d, last mip %g\n", sizeof(block), mifx_status_string(st), attribs.LightCount, attribs.PrefilteredCubeLastMip);
        PBRGBufferShadeAttribs sh;
        sh.pPostFXContext    = &postfx;
        sh.pFrameAttribsData = &block;
        sh.FrameAttribsSize  = sizeof(block);
        sh.MaxLightCount     = 2;
        ShadeGBuffer(sh); // no shared cube maps in this driver: logged, not fatal
    }

    // the shade with material layers: the per-material scalars come out of the renderer's own material block (PBRMaterialShaderInfo of a pipeline with ENABLE_ANISOTROPY and
    // ENABLE_IRIDESCENCE, two texture-attribute blocks: PBR_Structures.fxh:291-317) -- host only; the call itself is refused here (no G-buffer), like the others above
    {
        struct MaterialBlock
        {
            mifx_pbr_material_basic_attribs       Basic;
            mifx_pbr_material_anisotropy_attribs  Anisotropy;
            mifx_pbr_material_iridescence_attribs Iridescence;
            float                                 Textures[2][12]; // PBRMaterialTextureAttribs, 48 bytes each
        } material{};
        material.Basic.Workflow    = MIFX_PBR_WORKFLOW_METALLIC_ROUGHNESS;
        material.Anisotropy.Rotation = 0.5f;
        material.Iridescence.IOR     = 1.3f;
        mifx_pbr_layers layers{};
        mifx_pbr_material_basic_attribs basic{};
        const uint32_t    set = MIFX_PBR_LAYER_ANISOTROPY | MIFX_PBR_LAYER_IRIDESCENCE;
        const mifx_status st  = mifx_pbr_layers_from_material_info(&material, sizeof(material), set, 0, 2, &layers, &basic);
        std::printf("PBRMaterialShaderInfo (%zu bytes, anisotropy + iridescence, 2 texture blocks): %s, rotation %g, IOR %g\n", sizeof(material), mifx_status_string(st),
                    layers.anisotropy_rotation, layers.iridescence_ior);
        const mifx_status refused = mifx_pbr_shade_execute_layers(postfx.GetMifxContext(), nullptr, &layers, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
        std::printf("layered shade without a G-buffer: %s\n", mifx_status_string(refused));
    }

    const bool outputs = ssao.GetAmbientOcclusionSRV() != nullptr || ssr.GetSSRRadianceSRV() != nullptr || taa.GetAccumulatedFrameSRV() != nullptr ||
        bloom.GetBloomTextureSRV() != nullptr || dof.GetDepthOfFieldTextureSRV() != nullptr;
    std::printf("device: %s; outputs handed out: %s\n", haveDevice ? "yes" : "no", outputs ? "yes" : "no");
    // without a device nothing may have been handed out; with one, prepare has allocated the effect-owned planes
    return (!haveDevice && outputs) ? 1 : 0;
}
