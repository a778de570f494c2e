// TEST INFRASTRUCTURE ONLY.  The product's HOST code (diligentfx_amd/csrc/api_*.cpp, mifx_core.cpp -- the objects behind the C ABI: what is prepared, cleared, ping-ponged and
// launched in which order) linked WITHOUT its kernels and without a GPU: this file stands in for the two things below it --
//   * the HIP runtime: memory is host memory, streams and events do nothing (everything "executes" at once, in program order);
//   * the kernel launchers (mifx_host.h launch_*, defined in the .hip files): each hands its arguments to a callback, which the test installs (tests/cpu_product/device.py runs
//     the reference's own shader for the pass on the planes it was given: oracle/_ref).
// What this makes testable on the CPU: the product's sequencing against the executed reference host classes (oracle/refhost) over random sequences of frames.  What it is not: a CPU
// path of the product -- nothing in diligentfx_amd/ builds or loads it, and the library that ships fails without its HIP kernels and a device.
#include "mifx_host.h"
#include <cstdlib>
#include <cstring>

// ---------------------------------------------------------------------------------------------------------------- HIP runtime stand-in
// Streams and events execute nothing -- but what the host code asks of them is reported to an optional second callback (tests/cpu_product/order.py), which keeps a vector
// clock per stream and event and checks every launch's reads and writes for a happens-before edge to the conflicting accesses before it: the ORDER the host code imposes
// on its lanes is checked without a device (a missing wait is a race here even when the device run happens to come out right).
typedef void (*mifx_cpu_runtime_callback)(int op, const void* a, const void* b, unsigned long long n);
static mifx_cpu_runtime_callback g_runtime = nullptr;
enum { RT_STREAM_CREATE = 1, RT_STREAM_DESTROY, RT_EVENT_CREATE, RT_EVENT_DESTROY, RT_EVENT_RECORD, RT_STREAM_WAIT, RT_STREAM_SYNC, RT_EVENT_SYNC, RT_WRITE, RT_READ, RT_MALLOC, RT_FREE };
static void rt(int op, const void* a, const void* b = nullptr, unsigned long long n = 0) { if (g_runtime) g_runtime(op, a, b, n); }
extern "C" {
__attribute__((visibility("default"))) void mifx_cpu_set_runtime_callback(mifx_cpu_runtime_callback cb) { g_runtime = cb; }
hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipMalloc(void** p, size_t n) { *p = std::calloc(1, n ? n : 1); if (*p) rt(RT_MALLOC, *p, nullptr, n ? n : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void* p) { if (p) rt(RT_FREE, p); std::free(p); return hipSuccess; }
hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t st) { rt(RT_READ, s, st, n); rt(RT_WRITE, d, st, n); std::memmove(d, s, n); return hipSuccess; }
hipError_t hipMemcpy2DAsync(void* d, size_t dp, const void* s, size_t sp, size_t w, size_t h, hipMemcpyKind, hipStream_t st)
{
    rt(RT_READ, s, st, sp * h);
    rt(RT_WRITE, d, st, dp * h);
    for (size_t y = 0; y < h; ++y) std::memmove(static_cast<char*>(d) + y * dp, static_cast<const char*>(s) + y * sp, w);
    return hipSuccess;
}
hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t st) { rt(RT_WRITE, d, st, n); std::memset(d, v, n); return hipSuccess; }
hipError_t hipMemsetD16Async(hipDeviceptr_t d, unsigned short v, size_t count, hipStream_t st)
{
    rt(RT_WRITE, d, st, count * 2u);
    unsigned short* p = static_cast<unsigned short*>(d);
    for (size_t i = 0; i < count; ++i) p[i] = v;
    return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = reinterpret_cast<hipStream_t>(std::malloc(8)); rt(RT_STREAM_CREATE, *s); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { rt(RT_STREAM_DESTROY, s); std::free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { rt(RT_STREAM_SYNC, s); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { rt(RT_STREAM_WAIT, s, e); return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e) { *e = reinterpret_cast<hipEvent_t>(std::malloc(8)); rt(RT_EVENT_CREATE, *e); return hipSuccess; }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e) { rt(RT_EVENT_DESTROY, e); std::free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { rt(RT_EVENT_RECORD, e, s); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { rt(RT_EVENT_SYNC, e); return hipSuccess; }
hipError_t hipEventQuery(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }
const char* hipGetErrorString(hipError_t) { return "stand-in HIP runtime"; }

// ---------------------------------------------------------------------------------------------------------------- the launch callback
struct mifx_cpu_arg // one argument of a launcher, in declaration order (the stream is dropped)
{
    int         kind; // 0 image (by value: `img`), 1 struct passed by reference (p, bytes), 2 integer (i), 3 float (f), 4 raw pointer (p)
    mifx::Img   img;
    const void* p;
    size_t      bytes;
    long long   i;
    double      f;
};
struct mifx_cpu_call
{
    const char*  name; // the launcher's name without "launch_"
    int          count;
    mifx_cpu_arg arg[40];
    const void*  stream; // the stream the launcher was given (what tests/cpu_product/order.py orders the launch by)
};
static thread_local const void* t_stream = nullptr;
typedef int (*mifx_cpu_callback)(const mifx_cpu_call*);
static mifx_cpu_callback g_callback = nullptr;
__attribute__((visibility("default"))) void mifx_cpu_set_launch_callback(mifx_cpu_callback cb) { g_callback = cb; }
}

namespace mifx
{
static void push(mifx_cpu_call& c, const Img& v) { c.arg[c.count].kind = 0; c.arg[c.count].img = v; ++c.count; }
static void push(mifx_cpu_call& c, const Pyr& v) { mifx_cpu_arg& a = c.arg[c.count++]; a.kind = 1; a.p = &v; a.bytes = sizeof(v); }
template <class T> static typename std::enable_if<std::is_class<T>::value>::type push(mifx_cpu_call& c, const T& v) { mifx_cpu_arg& a = c.arg[c.count++]; a.kind = 1; a.p = &v; a.bytes = sizeof(T); }
template <class T> static typename std::enable_if<std::is_integral<T>::value || std::is_enum<T>::value>::type push(mifx_cpu_call& c, T v) { mifx_cpu_arg& a = c.arg[c.count++]; a.kind = 2; a.i = (long long)v; }
static void push(mifx_cpu_call& c, float v) { mifx_cpu_arg& a = c.arg[c.count++]; a.kind = 3; a.f = v; }
template <class T> static void push(mifx_cpu_call& c, T* v) { mifx_cpu_arg& a = c.arg[c.count++]; a.kind = 4; a.p = v; }
template <class... A> static mifx_status record(const char* name, const A&... a)
{
    if (g_callback == nullptr)
    {
        set_error("launch_%s: no launch callback installed (tests/cpu_product)", name);
        return MIFX_ERR_NOT_IMPLEMENTED;
    }
    mifx_cpu_call c{};
    c.name   = name;
    c.stream = t_stream;
    (push(c, a), ...);
    return static_cast<mifx_status>(g_callback(&c));
}

// two host helpers that live beside their kernels
bool bloom_tail_fits(const Img*, int) { return false; } // (the one-workgroup tail of the Bloom pyramid is a fusion of B2 / B3 launches: the wide kernels compute the same levels)
uint32_t native_texel_size(uint32_t fmt)
{
    switch (fmt)
    {
        case MIFX_NATIVE_FORMAT_R8_UNORM: return 1;
        case MIFX_NATIVE_FORMAT_R16_FLOAT: case MIFX_NATIVE_FORMAT_RG8_UNORM: case MIFX_NATIVE_FORMAT_R16_UNORM: return 2;
        case MIFX_NATIVE_FORMAT_R32_FLOAT: case MIFX_NATIVE_FORMAT_RG16_FLOAT: case MIFX_NATIVE_FORMAT_RGBA8_UNORM: case MIFX_NATIVE_FORMAT_RGBA8_UNORM_SRGB:
        case MIFX_NATIVE_FORMAT_RG16_UNORM: case MIFX_NATIVE_FORMAT_R11G11B10_FLOAT: return 4;
        case MIFX_NATIVE_FORMAT_RG32_FLOAT: case MIFX_NATIVE_FORMAT_RGBA16_FLOAT: case MIFX_NATIVE_FORMAT_RGBA16_UNORM: return 8;
        case MIFX_NATIVE_FORMAT_RGBA32_FLOAT: return 16;
        default: return 0;
    }
}

// ---------------------------------------------------------------------------------------------------------------- the launchers (generated from mifx_host.h by tests/cpu_product/gen_stubs.py)
mifx_status launch_fill_f32(hipStream_t s, Img plane, int floats_per_texel, float value)
{
    t_stream = s;
    return record("fill_f32", plane, floats_per_texel, value);
}
mifx_status launch_clear_texels(hipStream_t s, Img plane, int channels, bool halves, const float color[4])
{
    t_stream = s;
    return record("clear_texels", plane, channels, halves, color);
}
mifx_status launch_stream_copy(hipStream_t s, const void* src, void* dst, unsigned long long bytes)
{
    t_stream = s;
    return record("stream_copy", src, dst, bytes);
}
mifx_status launch_eval_math(hipStream_t s, unsigned op, const float* a, const float* b, float* out, unsigned long long n)
{
    t_stream = s;
    return record("eval_math", op, a, b, out, n);
}
mifx_status launch_tonemap(hipStream_t s, Img in, Img out, const mifx_tone_mapping_attribs& a, float ave_log_lum, uint32_t flags, const float* aveLum, bool packedIn)
{
    t_stream = s;
    return record("tonemap", in, out, a, ave_log_lum, flags, aveLum, packedIn);
}
mifx_status launch_tonemap_native(hipStream_t s, Img in, const mifx_native_image* ldr_out, const mifx_tone_mapping_attribs& a, float ave_log_lum, uint32_t flags, const float* aveLum, bool packedIn)
{
    t_stream = s;
    return record("tonemap_native", in, ldr_out, a, ave_log_lum, flags, aveLum, packedIn);
}
mifx_status launch_autoexposure(hipStream_t s, Img color, Img lowRes, float* average, float elapsedTime, int lightAdaptation, bool packedIn)
{
    t_stream = s;
    return record("autoexposure", color, lowRes, average, elapsedTime, lightAdaptation, packedIn);
}
mifx_status launch_autoexposure_rows(hipStream_t s, Img color, Img lowRes, int rowBegin, int rowEnd, bool packedIn)
{
    t_stream = s;
    return record("autoexposure_rows", color, lowRes, rowBegin, rowEnd, packedIn);
}
mifx_status launch_autoexposure_reduce(hipStream_t s, Img lowRes, float* average, float elapsedTime, int lightAdaptation)
{
    t_stream = s;
    return record("autoexposure_reduce", lowRes, average, elapsedTime, lightAdaptation);
}
mifx_status launch_postfx_prep(hipStream_t s, Img depth, Img motion, Img reproj, Img closest, const CamK& cur, const CamK& prev, const uint8_t* sobol, const uint8_t* tile, Img noiseXY, Img noiseZW, uint32_t frame, bool halfPrecisionDepth)
{
    t_stream = s;
    return record("postfx_prep", depth, motion, reproj, closest, cur, prev, sobol, tile, noiseXY, noiseZW, frame, halfPrecisionDepth);
}
mifx_status launch_depth16_copy(hipStream_t s, Img in, Img out)
{
    t_stream = s;
    return record("depth16_copy", in, out);
}
mifx_status launch_ssao_prefilter_pyramid(hipStream_t s, const Pyr& p, const Pyr& camz, const CamK& cam, const mifx_ssao_attribs& a, bool depth16)
{
    t_stream = s;
    return record("ssao_prefilter_pyramid", p, camz, cam, a, depth16);
}
mifx_status launch_ssao_compute_ao(hipStream_t s, const Pyr& depthPyr, const Pyr& camzPyr, Img normal, Img noiseZW, Img out, const CamK& cam, const mifx_ssao_attribs& a, bool halfResolution, bool halfPrecisionDepth)
{
    t_stream = s;
    return record("ssao_compute_ao", depthPyr, camzPyr, normal, noiseZW, out, cam, a, halfResolution, halfPrecisionDepth);
}
mifx_status launch_ssao_downsample_depth(hipStream_t s, Img depth, Img out)
{
    t_stream = s;
    return record("ssao_downsample_depth", depth, out);
}
mifx_status launch_ssao_depth_to_camz(hipStream_t s, Img depth, Img camz, const CamK& cam)
{
    t_stream = s;
    return record("ssao_depth_to_camz", depth, camz, cam);
}
mifx_status launch_ssao_bilateral_upsample(hipStream_t s, Img depth, Img occlusion, Img out, const CamK& cam)
{
    t_stream = s;
    return record("ssao_bilateral_upsample", depth, occlusion, out, cam);
}
mifx_status launch_ssao_temporal(hipStream_t s, Img currAO, Img prevAO, Img prevLen, Img reprojDepth, Img prevDepth, Img motion, Img outAO, Img outLen, const CamK& cur, const CamK& prev, const mifx_ssao_attribs& a, const SsaoResolve* resolve)
{
    t_stream = s;
    return record("ssao_temporal", currAO, prevAO, prevLen, reprojDepth, prevDepth, motion, outAO, outLen, cur, prev, a, resolve);
}
mifx_status launch_ssao_resolve_lists(hipStream_t s, const Pyr& aoPyr, const Pyr& depthPyr, Img histLen, Img camz, Img normal, Img rows5, const SsaoResolve& r, const CamK& cam, const mifx_ssao_attribs& a)
{
    t_stream = s;
    return record("ssao_resolve_lists", aoPyr, depthPyr, histLen, camz, normal, rows5, r, cam, a);
}
mifx_status launch_ssao_convolute_pyramids(hipStream_t s, const Pyr& ao, const Pyr& depth, bool depth16)
{
    t_stream = s;
    return record("ssao_convolute_pyramids", ao, depth, depth16);
}
mifx_status launch_ssao_resample(hipStream_t s, const Pyr& aoPyr, const Pyr& depthPyr, Img histLen, Img normal, Img out, const CamK& cam)
{
    t_stream = s;
    return record("ssao_resample", aoPyr, depthPyr, histLen, normal, out, cam);
}
mifx_status launch_ssao_spatial(hipStream_t s, Img occl, Img histLen, Img depth, Img camz, Img normal, Img out, Img historyOut, const CamK& cam, const mifx_ssao_attribs& a)
{
    t_stream = s;
    return record("ssao_spatial", occl, histLen, depth, camz, normal, out, historyOut, cam, a);
}
mifx_status launch_pbr_shade(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_spec, int row_begin, int row_end, bool reversedDepth, const mifx_pbr_shadows* shadows, const SsrMaskOut* ssrMask)
{
    t_stream = s;
    return record("pbr_shade", iblApron, g, camera, a, ibl, background, out_radiance, out_spec, row_begin, row_end, reversedDepth, shadows, ssrMask);
}
mifx_status launch_pbr_shade_layers(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_pbr_layers& layers, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_spec, int row_begin, int row_end, bool reversedDepth, const mifx_pbr_shadows* shadows, const LayeredHitFetch* hit)
{
    t_stream = s;
    return record("pbr_shade_layers", iblApron, g, layers, camera, a, ibl, background, out_radiance, out_spec, row_begin, row_end, reversedDepth, shadows, hit);
}
mifx_status launch_pbr_hit_fetch(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], Img rays, Img hitCoords, const mifx_image2d* radiance, int shadedBegin, int shadedEnd, bool reversedDepth)
{
    t_stream = s;
    return record("pbr_hit_fetch", iblApron, g, camera, a, ibl, background, rays, hitCoords, radiance, shadedBegin, shadedEnd, reversedDepth);
}
mifx_status launch_pbr_shade_native(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer_native* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], const mifx_native_image* out_radiance, const mifx_native_image* out_spec, bool reversedDepth)
{
    t_stream = s;
    return record("pbr_shade_native", iblApron, g, camera, a, ibl, background, out_radiance, out_spec, reversedDepth);
}
mifx_status launch_composite(hipStream_t s, const mifx_composite_attribs& a, const mifx_image2d* out, int row_begin, int row_end, const SsrCleanupIn* r7)
{
    t_stream = s;
    return record("composite", a, out, row_begin, row_end, r7);
}
mifx_status launch_specgloss_material(hipStream_t s, Img baseColor, Img physicalDesc, Img out)
{
    t_stream = s;
    return record("specgloss_material", baseColor, physicalDesc, out);
}
mifx_status launch_bloom_prefilter(hipStream_t s, Img in, Img out, const mifx_bloom_attribs& a, bool packedInput)
{
    t_stream = s;
    return record("bloom_prefilter", in, out, a, packedInput);
}
mifx_status launch_bloom_downsample(hipStream_t s, Img in, Img out)
{
    t_stream = s;
    return record("bloom_downsample", in, out);
}
mifx_status launch_bloom_upsample(hipStream_t s, Img input, Img down, Img out, const mifx_bloom_attribs& a, bool final_pass, bool packedInput)
{
    t_stream = s;
    return record("bloom_upsample", input, down, out, a, final_pass, packedInput);
}
mifx_status launch_bloom_tail(hipStream_t s, const Img* down, const Img* up, int count)
{
    t_stream = s;
    return record("bloom_tail", down, up, count);
}
mifx_status launch_bloom_final_tonemap(hipStream_t s, Img input, Img down, Img out, Img ldr, const mifx_bloom_attribs& a, const mifx_tone_mapping_attribs& attr, float ave_log_lum, uint32_t flags, bool writeBloomOutput, bool packedInput)
{
    t_stream = s;
    return record("bloom_final_tonemap", input, down, out, ldr, a, attr, ave_log_lum, flags, writeBloomOutput, packedInput);
}
mifx_status launch_taa(hipStream_t s, Img currColor, Img prevColor, Img motion, Img reprojDepth, Img prevDepth, Img out, const CamK& cur, const CamK& prev, const mifx_taa_attribs& a, uint32_t flags, const TaaFusedComposite* fused)
{
    t_stream = s;
    return record("taa", currColor, prevColor, motion, reprojDepth, prevDepth, out, cur, prev, a, flags, fused);
}
mifx_status launch_dof_coc(hipStream_t s, Img depth, Img out, const mifx_camera_attribs& cam, float maxCoC)
{
    t_stream = s;
    return record("dof_coc", depth, out, cam, maxCoC);
}
mifx_status launch_dof_temporal_coc(hipStream_t s, Img curr, Img prev, Img motion, Img out, const mifx_camera_attribs& cam, float stability)
{
    t_stream = s;
    return record("dof_temporal_coc", curr, prev, motion, out, cam, stability);
}
mifx_status launch_dof_dilation(hipStream_t s, Img coc, const Img levels[3])
{
    t_stream = s;
    return record("dof_dilation", coc, levels);
}
mifx_status launch_dof_blur(hipStream_t s, Img in, Img out, const float weights[13])
{
    t_stream = s;
    return record("dof_blur", in, out, weights);
}
mifx_status launch_dof_prefilter(hipStream_t s, Img color, Img coc, Img dilation, Img outNear, Img outFar)
{
    t_stream = s;
    return record("dof_prefilter", color, coc, dilation, outNear, outFar);
}
mifx_status launch_dof_bokeh_gather(hipStream_t s, Img nearTex, Img farTex, Img radiance, Img outNear, Img outFar, const float* kernel, int sampleCount, float maxCoC, float aspect, bool karis)
{
    t_stream = s;
    return record("dof_bokeh_gather", nearTex, farTex, radiance, outNear, outFar, kernel, sampleCount, maxCoC, aspect, karis);
}
mifx_status launch_dof_bokeh_fill(hipStream_t s, Img nearTex, Img farTex, Img outNear, Img outFar, const float* kernel, int sampleCount, float maxCoC, float aspect)
{
    t_stream = s;
    return record("dof_bokeh_fill", nearTex, farTex, outNear, outFar, kernel, sampleCount, maxCoC, aspect);
}
mifx_status launch_dof_postfilter(hipStream_t s, Img nearTex, Img farTex, Img outNear, Img outFar)
{
    t_stream = s;
    return record("dof_postfilter", nearTex, farTex, outNear, outFar);
}
mifx_status launch_dof_combine(hipStream_t s, Img color, Img nearTex, Img farTex, Img out, float alpha)
{
    t_stream = s;
    return record("dof_combine", color, nearTex, farTex, out, alpha);
}
mifx_status launch_image_import(hipStream_t s, const mifx_native_image* src, const mifx_image2d* dst)
{
    t_stream = s;
    return record("image_import", src, dst);
}
mifx_status launch_image_export(hipStream_t s, const mifx_image2d* src, const mifx_native_image* dst)
{
    t_stream = s;
    return record("image_export", src, dst);
}
mifx_status launch_ssr_hiz_pyramid(hipStream_t s, const Pyr& p, Img level0Copy, bool reversedDepth)
{
    t_stream = s;
    return record("ssr_hiz_pyramid", p, level0Copy, reversedDepth);
}
mifx_status launch_ssr_mask_roughness(hipStream_t s, Img material, Img depth, Img roughness, Img mask, const mifx_ssr_attribs& a, bool reversedDepth)
{
    t_stream = s;
    return record("ssr_mask_roughness", material, depth, roughness, mask, a, reversedDepth);
}
mifx_status launch_ssr_intersection(hipStream_t s, Img radiance, Img normal, Img roughness, Img noiseXY, const HizSlab& hiz, Img mask, Img motion, Img outSpec, Img outDirPdf, const CamK& cam, const mifx_ssr_attribs& a, bool previousFrame, bool halfResolution, Img hitCoords, int localBegin, int localEnd)
{
    t_stream = s;
    return record("ssr_intersection", radiance, normal, roughness, noiseXY, hiz, mask, motion, outSpec, outDirPdf, cam, a, previousFrame, halfResolution, hitCoords, localBegin, localEnd);
}
mifx_status launch_ssr_downsampled_mask(hipStream_t s, Img roughness, Img depth, Img mask, const mifx_ssr_attribs& a, bool reversedDepth)
{
    t_stream = s;
    return record("ssr_downsampled_mask", roughness, depth, mask, a, reversedDepth);
}
mifx_status launch_ssr_spatial(hipStream_t s, Img roughness, Img normal, Img depth, Img dirPdf, Img spec, Img mask, Img outRad, Img outVar, Img outDepth, const CamK& cam, const mifx_ssr_attribs& a, bool halfResolution)
{
    t_stream = s;
    return record("ssr_spatial", roughness, normal, depth, dirPdf, spec, mask, outRad, outVar, outDepth, cam, a, halfResolution);
}
mifx_status launch_ssr_temporal(hipStream_t s, Img motion, Img hitDepth, Img reprojDepth, Img currRad, Img currVar, Img prevDepth, Img prevRad, Img prevVar, Img mask, Img outRad, Img outVar, const CamK& cur, const CamK& prev, const mifx_ssr_attribs& a)
{
    t_stream = s;
    return record("ssr_temporal", motion, hitDepth, reprojDepth, currRad, currVar, prevDepth, prevRad, prevVar, mask, outRad, outVar, cur, prev, a);
}
mifx_status launch_ssr_bilateral(hipStream_t s, Img normal, const SsrCleanupIn& in, Img out, const CamK& cam)
{
    t_stream = s;
    return record("ssr_bilateral", normal, in, out, cam);
}
mifx_status launch_ibl_brdf_lut(hipStream_t s, Img out, uint32_t num_samples)
{
    t_stream = s;
    return record("ibl_brdf_lut", out, num_samples);
}
mifx_status launch_ibl_prefilter(hipStream_t s, const mifx_cubemap* env, const mifx_spheremap* sphere, void* out, uint32_t out_size, float roughness, uint32_t num_samples)
{
    t_stream = s;
    return record("ibl_prefilter", env, sphere, out, out_size, roughness, num_samples);
}
mifx_status launch_ibl_irradiance(hipStream_t s, const mifx_cubemap* env, const mifx_spheremap* sphere, void* out, uint32_t out_size, uint32_t num_samples)
{
    t_stream = s;
    return record("ibl_irradiance", env, sphere, out, out_size, num_samples);
}
mifx_status launch_envmap(hipStream_t s, const mifx_envmap_render_attribs& a, const mifx_tone_mapping_attribs& tm, const mifx_camera_attribs& cam, const mifx_camera_attribs& prev, Img depth, Img color, Img motion)
{
    t_stream = s;
    return record("envmap", a, tm, cam, prev, depth, color, motion);
}
} // namespace mifx
