// adapter_frame.cpp -- REAL frames through the adapters: the reference's class interfaces (Diligent::PostFXContext, ScreenSpaceAmbientOcclusion, ScreenSpaceReflection,
// TemporalAntiAliasing, Bloom as declared in mifx_effect_adapters.hpp) driven from C++ on device planes, in the order and with the protocol of HnPostProcessTask
// (Prepare :671-682, Execute :788-918).  The interop a host application provides (mifx_interop.hpp) is the simplest possible one here: a texture view IS a pitched HIP
// allocation.  tests/test_adapter_example.py feeds the program the inputs of a few frames, runs the same frames through the ctypes mirror of the C ABI and compares
// every effect output bit for bit: nothing in the path from the reference's interface to the kernels is Python.
//
//   adapter_frame <input file> <output file>          g++ -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include ... -lmifx -lamdhip64
// input:  uint32 W, H, frames; 256 + 131072 bytes of blue-noise tables; then per frame  uint32 index, 576 + 576 bytes of cameras, the planes depth (W*H floats),
//         prev_depth, motion (x2), normal (x4), material (x4), colour (x4)
// output: per frame the planes AO (W*H floats), SSR (x4), TAA (x4), Bloom (x4), closest motion (x2), reprojected depth
#include <hip/hip_runtime_api.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "mifx_effect_adapters.hpp"

namespace Diligent
{
struct ITextureView // the application's view of a texture: a pitched device plane
{
    mifx_image2d img;
};
namespace NoiseBuffers
{
unsigned char Sobol_256d_storage[256];
unsigned char ScramblingTile_storage[128 * 128 * 8];
} // namespace NoiseBuffers
mifx_image2d GetMifxImage(ITextureView* pView, uint32_t Format)
{
    mifx_image2d img{};
    if (pView != nullptr) img = pView->img;
    if (pView != nullptr && img.format != Format) std::fprintf(stderr, "GetMifxImage: the view holds format %u, the effect expects %u\n", img.format, Format);
    return img;
}
ITextureView* WrapMifxImage(const mifx_image2d& Image)
{
    static ITextureView views[64];
    static unsigned     next = 0;
    ITextureView&       v    = views[next++ % 64u];
    v.img                    = Image;
    return &v;
}
void* GetMifxStream(IDeviceContext*) { return nullptr; }
bool  GetMifxCubemap(ITextureView*, mifx_cubemap&) { return false; }
} // namespace Diligent

using namespace Diligent;

namespace
{
#define HIP_OK(call)                                                                                 \
    do {                                                                                             \
        const hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #call, hipGetErrorString(e_)); std::exit(2); } \
    } while (0)

struct DevicePlane
{
    ITextureView view{};
    uint32_t     channels = 0;
    void alloc(uint32_t w, uint32_t h, uint32_t c, uint32_t fmt)
    {
        channels      = c;
        size_t pitch  = (size_t(w) * c * 4u + 255u) & ~size_t(255); // a row pitch the application chose: the library takes any
        void*  p      = nullptr;
        HIP_OK(hipMalloc(&p, pitch * h));
        view.img = mifx_image2d{p, w, h, uint32_t(pitch), fmt};
    }
    void upload(const float* src)
    {
        HIP_OK(hipMemcpy2D(view.img.data, view.img.pitch_bytes, src, size_t(view.img.width) * channels * 4u, size_t(view.img.width) * channels * 4u, view.img.height, hipMemcpyHostToDevice));
    }
};
void download(ITextureView* v, uint32_t channels, std::vector<float>& dst)
{
    dst.assign(size_t(v->img.width) * v->img.height * channels, 0.0f);
    HIP_OK(hipMemcpy2D(dst.data(), size_t(v->img.width) * channels * 4u, v->img.data, v->img.pitch_bytes, size_t(v->img.width) * channels * 4u, v->img.height, hipMemcpyDeviceToHost));
}
template <class T> T read(std::FILE* f)
{
    T v{};
    if (std::fread(&v, sizeof(T), 1, f) != 1) { std::fprintf(stderr, "short input file\n"); std::exit(2); }
    return v;
}
void read_floats(std::FILE* f, std::vector<float>& v, size_t n)
{
    v.resize(n);
    if (std::fread(v.data(), sizeof(float), n, f) != n) { std::fprintf(stderr, "short input file\n"); std::exit(2); }
}
} // namespace

// the tables PostFXContext.cpp:155-190 uploads: the stand-in header declares them const; this driver fills them from the input file before the context is created
namespace Diligent { namespace NoiseBuffers {
extern const unsigned char Sobol_256d[256]                __attribute__((alias("_ZN8Diligent12NoiseBuffers18Sobol_256d_storageE")));
extern const unsigned char ScramblingTile[128 * 128 * 8] __attribute__((alias("_ZN8Diligent12NoiseBuffers22ScramblingTile_storageE")));
}} // namespace Diligent::NoiseBuffers

int main(int argc, char** argv)
{
    if (argc != 3) { std::fprintf(stderr, "usage: adapter_frame <input> <output>\n"); return 2; }
    std::FILE* in = std::fopen(argv[1], "rb");
    std::FILE* out = std::fopen(argv[2], "wb");
    if (!in || !out) { std::fprintf(stderr, "cannot open the files\n"); return 2; }
    const uint32_t W = read<uint32_t>(in), H = read<uint32_t>(in), frames = read<uint32_t>(in);
    if (std::fread(NoiseBuffers::Sobol_256d_storage, 1, 256, in) != 256 || std::fread(NoiseBuffers::ScramblingTile_storage, 1, 128 * 128 * 8, in) != 128 * 128 * 8) return 2;

    PostFXContext               postfx(nullptr, PostFXContext::CreateInfo{});
    ScreenSpaceAmbientOcclusion ssao(nullptr, {});
    ScreenSpaceReflection       ssr(nullptr, {});
    TemporalAntiAliasing        taa(nullptr, {});
    Bloom                       bloom(nullptr, {});
    if (postfx.GetMifxContext() == nullptr) { std::fprintf(stderr, "no device\n"); return 3; }

    DevicePlane depth, prevDepth, motion, normal, material, color;
    depth.alloc(W, H, 1, MIFX_FORMAT_F32); prevDepth.alloc(W, H, 1, MIFX_FORMAT_F32); motion.alloc(W, H, 2, MIFX_FORMAT_F32X2);
    normal.alloc(W, H, 4, MIFX_FORMAT_F32X4); material.alloc(W, H, 4, MIFX_FORMAT_F32X4); color.alloc(W, H, 4, MIFX_FORMAT_F32X4);

    // attribute blocks with the reference's defaults (Shaders/PostProcess/*/public/*Structures.fxh DEFAULT_VALUE): written as the floats / ints the blocks hold
    mifx_ssao_attribs  ssaoA{};
    mifx_ssr_attribs   ssrA{};
    mifx_taa_attribs   taaA{};
    mifx_bloom_attribs bloomA{};
    if (std::fread(&ssaoA, sizeof(ssaoA), 1, in) != 1 || std::fread(&ssrA, sizeof(ssrA), 1, in) != 1 || std::fread(&taaA, sizeof(taaA), 1, in) != 1 || std::fread(&bloomA, sizeof(bloomA), 1, in) != 1) return 2;
    HLSL::ScreenSpaceAmbientOcclusionAttribs ssaoH; std::memcpy(&ssaoH, &ssaoA, sizeof(ssaoH));
    HLSL::ScreenSpaceReflectionAttribs       ssrH;  std::memcpy(&ssrH, &ssrA, sizeof(ssrH));
    HLSL::TemporalAntiAliasingAttribs        taaH;  std::memcpy(&taaH, &taaA, sizeof(taaH));
    HLSL::BloomAttribs                       bloomH; std::memcpy(&bloomH, &bloomA, sizeof(bloomH));

    std::vector<float> buf;
    for (uint32_t n = 0; n < frames; ++n)
    {
        const uint32_t index = read<uint32_t>(in);
        HLSL::CameraAttribs cam = read<HLSL::CameraAttribs>(in), prevCam = read<HLSL::CameraAttribs>(in);
        read_floats(in, buf, size_t(W) * H); depth.upload(buf.data());
        read_floats(in, buf, size_t(W) * H); prevDepth.upload(buf.data());
        read_floats(in, buf, size_t(W) * H * 2); motion.upload(buf.data());
        read_floats(in, buf, size_t(W) * H * 4); normal.upload(buf.data());
        read_floats(in, buf, size_t(W) * H * 4); material.upload(buf.data());
        read_floats(in, buf, size_t(W) * H * 4); color.upload(buf.data());

        // HnPostProcessTask::Prepare
        PostFXContext::FrameDesc fd;
        fd.Index = index; fd.Width = fd.OutputWidth = W; fd.Height = fd.OutputHeight = H;
        postfx.PrepareResources(nullptr, fd, PostFXContext::FEATURE_FLAG_NONE);
        ssao.PrepareResources(nullptr, nullptr, &postfx, ScreenSpaceAmbientOcclusion::FEATURE_FLAG_NONE);
        ssr.PrepareResources(nullptr, nullptr, &postfx, ScreenSpaceReflection::FEATURE_FLAG_NONE);
        taa.PrepareResources(nullptr, nullptr, &postfx, TemporalAntiAliasing::FEATURE_FLAG_BICUBIC_FILTER);
        bloom.PrepareResources(nullptr, nullptr, &postfx, Bloom::FEATURE_FLAG_NONE);
        // HnPostProcessTask::Execute
        PostFXContext::RenderAttributes pra;
        pra.pCurrDepthBufferSRV = &depth.view; pra.pPrevDepthBufferSRV = &prevDepth.view; pra.pMotionVectorsSRV = &motion.view;
        pra.pCurrCamera = &cam; pra.pPrevCamera = &prevCam;
        postfx.Execute(pra);
        ScreenSpaceReflection::RenderAttributes rra;
        rra.pPostFXContext = &postfx; rra.pColorBufferSRV = &color.view; rra.pDepthBufferSRV = &depth.view; rra.pNormalBufferSRV = &normal.view;
        rra.pMaterialBufferSRV = &material.view; rra.pMotionVectorsSRV = &motion.view; rra.pSSRAttribs = &ssrH;
        ssr.Execute(rra);
        ScreenSpaceAmbientOcclusion::RenderAttributes sra;
        sra.pPostFXContext = &postfx; sra.pDepthBufferSRV = &depth.view; sra.pNormalBufferSRV = &normal.view; sra.pSSAOAttribs = &ssaoH;
        ssao.Execute(sra);
        TemporalAntiAliasing::RenderAttributes tra;
        tra.pPostFXContext = &postfx; tra.pColorBufferSRV = &color.view; tra.pTAAAttribs = &taaH;
        taa.Execute(tra);
        Bloom::RenderAttributes bra;
        bra.pPostFXContext = &postfx; bra.pColorBufferSRV = taa.GetAccumulatedFrameSRV(); bra.pBloomAttribs = &bloomH;
        bloom.Execute(bra);
        HIP_OK(hipDeviceSynchronize());
        struct { ITextureView* v; uint32_t c; } outs[] = {{ssao.GetAmbientOcclusionSRV(), 1}, {ssr.GetSSRRadianceSRV(), 4}, {taa.GetAccumulatedFrameSRV(), 4}, {bloom.GetBloomTextureSRV(), 4},
                                                           {postfx.GetClosestMotionVectors(), 2}, {postfx.GetReprojectedDepth(), 1}};
        for (auto& o : outs)
        {
            if (o.v == nullptr) { std::fprintf(stderr, "an effect handed out no output\n"); return 4; }
            download(o.v, o.c, buf);
            std::fwrite(buf.data(), sizeof(float), buf.size(), out);
        }
    }
    std::fclose(in);
    std::fclose(out);
    std::printf("adapter_frame: %u frames of %ux%u through Diligent::PostFXContext / ScreenSpaceReflection / ScreenSpaceAmbientOcclusion / TemporalAntiAliasing / Bloom\n", frames, W, H);
    return 0;
}
