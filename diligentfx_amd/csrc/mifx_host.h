// mifx_host.h -- host-side plumbing of libmifx: status/error handling, owned device planes, kernel launcher
// declarations.  Host objects mirror the reference's effect classes (PrepareResources / Execute / Get*SRV)
// without any render-device abstraction: a "texture" is a pitched HBM plane, a "pass" is a kernel launch on the
// context's HIP stream.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <utility>
#include <vector>

#include "mifx.h"
#include "mifx_device.h"
#include "mifx_ssr_cleanup.h"

namespace mifx
{
void set_error(const char* fmt, ...);
// experiment knob of the two gather kernels: bytes of unused dynamic LDS per workgroup from the environment (0 when unset; clamped to 64 KB - what the kernels use)
unsigned occupancy_pad_from_env(const char* name);

// rocTX ranges named after the reference's ScopedDebugGroup markers ("ScreenSpaceAmbientOcclusion", "ComputeAmbientOcclusion", ...:
// ScreenSpaceAmbientOcclusion.cpp:363,976, ScreenSpaceReflection.cpp:315, Bloom.cpp:296, ...), so that a rocprofv3 --marker-trace timeline reads like a
// RenderDoc / PIX capture of the reference.  Off unless MIFX_ROCTX=1 (or mifx_set_markers): a disabled range is one predictable branch.  libroctx64 is
// loaded with dlopen the first time a range is pushed.
struct MifxRange
{
    bool on;
    explicit MifxRange(const char* name);
    void end(); // closes the range now (the destructor then does nothing)
    ~MifxRange() { end(); }
    MifxRange(const MifxRange&) = delete;
    MifxRange& operator=(const MifxRange&) = delete;
};
#define MIFX_RANGE_CAT2(a, b) a##b
#define MIFX_RANGE_CAT(a, b) MIFX_RANGE_CAT2(a, b)
#define MIFX_RANGE(name) ::mifx::MifxRange MIFX_RANGE_CAT(mifx_range_, __LINE__)(name)
void set_markers(int enable);

#define MIFX_HIP_CHECK(expr)                                                                     \
    do                                                                                           \
    {                                                                                            \
        hipError_t _e = (expr);                                                                  \
        if (_e != hipSuccess)                                                                    \
        {                                                                                        \
            ::mifx::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
            return MIFX_ERR_HIP;                                                                 \
        }                                                                                        \
    } while (0)

#define MIFX_CHECK(st)                  \
    do                                  \
    {                                   \
        mifx_status _s = (st);          \
        if (_s < 0) return _s;          \
    } while (0)

#define MIFX_REQUIRE(cond, ...)                   \
    do                                            \
    {                                             \
        if (!(cond))                              \
        {                                         \
            ::mifx::set_error(__VA_ARGS__);       \
            return MIFX_ERR_INVALID_ARG;          \
        }                                         \
    } while (0)

inline uint32_t texel_size(uint32_t fmt)
{
    switch (fmt)
    {
        case MIFX_FORMAT_F32: return 4u;
        case MIFX_FORMAT_F32X2: return 8u;
        case MIFX_FORMAT_F32X4: return 16u;
        case MIFX_FORMAT_F16X4: return 8u;
        case MIFX_FORMAT_U8: return 1u;
        case MIFX_FORMAT_F16: return 2u;
        case MIFX_FORMAT_F16X2: return 4u;
        case MIFX_FORMAT_R11G11B10: return 4u;
        case MIFX_FORMAT_U16: return 2u;
        default: return 0u;
    }
}
// What a plane holds, for the planes whose storage type depends on the build (mifx_device.h: ao_t, hl_t, rough_t, var_t, cm_t, bloom_t, coc_t, dil_t): named at every
// Plane::alloc / to_img of such a plane and resolved by storage_format().
enum : uint32_t
{
    MIFX_PLANE_AO            = 0x101, // ambient occlusion
    MIFX_PLANE_HISTORY_LEN   = 0x102, // SSAO history length
    MIFX_PLANE_ROUGHNESS     = 0x103, // SSR roughness
    MIFX_PLANE_VARIANCE      = 0x104, // SSR variance / resolved depth
    MIFX_PLANE_CLOSEST_MOTION = 0x105,
    MIFX_PLANE_BLOOM         = 0x106, // Bloom pyramid levels and output
    MIFX_PLANE_MASK          = 0x107, // SSR reflection mask
    MIFX_PLANE_COC           = 0x108, // depth of field: signed circle of confusion (and its temporal history)
    MIFX_PLANE_COC_DILATION  = 0x109  // depth of field: dilated / blurred near-field circle of confusion
};
// The library's sources say MIFX_FORMAT_F32X4 for "the 4-channel texel"; the native-storage build (-DMIFX_STORAGE_H4) allocates, demands and hands out
// MIFX_FORMAT_F16X4 in its place (mifx_device.h: GlobalAccess<v4>).
inline uint32_t storage_format(uint32_t fmt)
{
    switch (fmt)
    {
#ifdef MIFX_STORAGE_H4
        case MIFX_FORMAT_F32X4: return MIFX_FORMAT_F16X4;
        case MIFX_PLANE_AO: case MIFX_PLANE_ROUGHNESS: case MIFX_PLANE_MASK: return MIFX_FORMAT_U8;
        case MIFX_PLANE_HISTORY_LEN: case MIFX_PLANE_VARIANCE: case MIFX_PLANE_COC: return MIFX_FORMAT_F16;
        case MIFX_PLANE_COC_DILATION: return MIFX_FORMAT_U16;
        case MIFX_PLANE_CLOSEST_MOTION: return MIFX_FORMAT_F16X2;
        case MIFX_PLANE_BLOOM: return MIFX_FORMAT_R11G11B10;
#else
        case MIFX_PLANE_AO: case MIFX_PLANE_ROUGHNESS: case MIFX_PLANE_HISTORY_LEN: case MIFX_PLANE_VARIANCE: case MIFX_PLANE_MASK: case MIFX_PLANE_COC:
        case MIFX_PLANE_COC_DILATION: return MIFX_FORMAT_F32;
        case MIFX_PLANE_CLOSEST_MOTION: return MIFX_FORMAT_F32X2;
        case MIFX_PLANE_BLOOM: return MIFX_FORMAT_F32X4;
#endif
        default: return fmt;
    }
}

// validates a borrowed image and converts it into a kernel view
mifx_status to_img(const mifx_image2d* im, uint32_t fmt, const char* what, Img& out);
mifx_status to_img_wh(const mifx_image2d* im, uint32_t fmt, uint32_t w, uint32_t h, const char* what, Img& out);

CamK make_camk(const mifx_camera_attribs& c, bool reversedDepth = false);

// an owned pitched plane in HBM (row pitch aligned to 256 B)
struct Plane
{
    void*    data = nullptr;
    uint32_t w = 0, h = 0, pitch = 0, fmt = 0;
    size_t   bytes = 0;
    bool     owned = true; // false: a view into memory owned by someone else (attach)
    Plane() = default;
    Plane(const Plane&) = delete;
    Plane& operator=(const Plane&) = delete;
    ~Plane() { release(); }
    mifx_status alloc(uint32_t width, uint32_t height, uint32_t format);
    void        release();
    void        attach(void* p, uint32_t width, uint32_t height, uint32_t pitch_bytes, uint32_t format) // non-owning view
    {
        release();
        data = p; w = width; h = height; pitch = pitch_bytes; fmt = format; bytes = size_t(pitch_bytes) * height; owned = false;
    }
    Img         view() const { return Img{static_cast<unsigned char*>(data), int(w), int(h), int(pitch), 0, 0}; } // y0 = yn = 0: all rows
    mifx_image2d desc() const { return mifx_image2d{data, w, h, pitch, fmt}; }
    mifx_status fill(hipStream_t s, float value) const; // every float of the plane := value
    // the two planes trade their memory (and geometry): how the chain double-buffers a plane an effect object owns without the effect knowing (api_chain.cpp, mode 4)
    void swap(Plane& o)
    {
        std::swap(data, o.data); std::swap(w, o.w); std::swap(h, o.h); std::swap(pitch, o.pitch); std::swap(fmt, o.fmt); std::swap(bytes, o.bytes); std::swap(owned, o.owned);
    }
    mifx_status alloc_like(const Plane& o) { return o.data ? alloc(o.w, o.h, o.fmt) : (release(), MIFX_OK); } // (o.fmt is a storage format already: storage_format() maps it to itself)
};

// The SSR depth hierarchy in one allocation (level 0 = a copy of the depth buffer, as in the reference, ScreenSpaceReflection.cpp:789-806):
// the ray march addresses it with a 32-bit offset from one base pointer
struct HizSlab
{
    const unsigned char* base;
    uint32_t offset[8], pitch[8], w[8], h[8];
    int      levels;
    uint32_t bytes; // size of the allocation (the range of the buffer resource the march reads it through)
    // Round 6: level 0 read where it lies -- the caller's depth plane -- through a second buffer resource, no copy into the slab (base0 != null: offset[0] is then 0 and
    // pitch[0] the plane's own pitch).  The copy is 86 % of the hierarchy pass' bytes; a row band of a sharded frame builds the WHOLE hierarchy on every rank.
    const unsigned char* base0;
    uint32_t             bytes0;
};

// grow-only device buffer for per-call working data (stream-ordered reuse; growing frees the old block, which waits for the device)
struct DeviceScratch
{
    void*  data  = nullptr;
    size_t bytes = 0;
    DeviceScratch() = default;
    DeviceScratch(const DeviceScratch&) = delete;
    DeviceScratch& operator=(const DeviceScratch&) = delete;
    ~DeviceScratch() { if (data) (void)hipFree(data); }
    mifx_status reserve(size_t n)
    {
        if (n <= bytes) return MIFX_OK;
        if (data) (void)hipFree(data);
        data = nullptr; bytes = 0;
        MIFX_HIP_CHECK(hipMalloc(&data, n));
        bytes = n;
        return MIFX_OK;
    }
    void swap(DeviceScratch& o) { std::swap(data, o.data); std::swap(bytes, o.bytes); }
    // Gives the block up WITHOUT freeing it: hipFree waits for the device, and a block that kernels which will never finish still touch (an exchange whose peer never
    // answers, on a transport without ncclCommAbort) would make that wait endless.
    void abandon() { data = nullptr; bytes = 0; }
};

// The shade's working copy of the IBL cube maps (one-texel apron per face, pbr.hip) and the promise under which it may be kept from call to call
// (mifx_postfx_set_static_ibl): `key` = the addresses and sizes of the maps the copy was made from.
struct IblApronCache
{
    DeviceScratch scratch;
    bool          keep  = false; // the caller declared the maps static
    bool          valid = false;
    const void*   key[26] = {};  // irradiance mip 0, prefiltered mips, sizes
};

inline dim3 tiled_grid(int w, int h) { return dim3((w + 31) / 32, (h + 7) / 8, 1); } // for kernels using tiled_xy()
inline dim3 grid2d(int w, int h, dim3 block) { return dim3((w + block.x - 1) / block.x, (h + block.y - 1) / block.y, 1); }
// launch shapes over the row window of the image a kernel writes (Img::y0 / yn; the whole image by default)
inline int  window_rows(const Img& out) { return (out.yn ? out.y0 + out.yn : out.h) - out.y0; }
inline dim3 tiled_grid(const Img& out) { return tiled_grid(out.w, window_rows(out)); }
inline dim3 grid2d(const Img& out, dim3 block) { return grid2d(out.w, window_rows(out), block); }
// number of levels (<= 4, <= remaining) that one launch of pyramid_reduce_levels (mifx_pyramid.h) can produce from a w x h source: every source on the way must have even dimensions
inline int pyramid_fusable_levels(int w, int h, int remaining)
{
    int n = 0;
    while (n < 4 && n < remaining && ((w >> n) & 1) == 0 && ((h >> n) & 1) == 0 && (w >> n) >= 2 && (h >> n) >= 2) ++n;
    return n;
}
inline Img  rows_of(Img im, int y0, int y1) // the same plane restricted to rows [y0, y1) (clipped)
{
    y0 = y0 < 0 ? 0 : y0;
    y1 = y1 > im.h ? im.h : y1;
    im.y0 = y0;
    im.yn = y1 > y0 ? y1 - y0 : 0;
    if (im.yn == 0) { im.y0 = 0; im.yn = 0; } // (an empty window is never launched; callers check)
    return im;
}

// ------------------------------------------------------------------------------------------------ kernel launchers (one per reference pass or fused group)
mifx_status launch_fill_f32(hipStream_t s, Img plane, int floats_per_texel, float value);
mifx_status launch_clear_texels(hipStream_t s, Img plane, int channels, bool halves, const float color[4]);
mifx_status launch_stream_copy(hipStream_t s, const void* src, void* dst, unsigned long long bytes);
mifx_status launch_eval_math(hipStream_t s, unsigned op, const float* a, const float* b, float* out, unsigned long long n);
// (packedIn: `in` is Bloom's R11G11B10_FLOAT output plane -- native-storage build; see to_img_hdr)
mifx_status launch_tonemap(hipStream_t s, Img in, Img out, const mifx_tone_mapping_attribs& a, float ave_log_lum, uint32_t flags, const float* aveLum = nullptr, bool packedIn = false);
mifx_status launch_tonemap_native(hipStream_t s, Img in, const mifx_native_image* ldr_out, const mifx_tone_mapping_attribs& a, float ave_log_lum, uint32_t flags,
                                  const float* aveLum = nullptr, bool packedIn = false);
// An HDR frame handed to the tone map / the auto exposure: the 4-channel colour texel of the build, or (native-storage build) MIFX_FORMAT_R11G11B10 -- what mifx_bloom_get_output
// hands out there, Bloom's output target in the reference's own format (Bloom.cpp:137).  packed = 1 for the latter.
mifx_status to_img_hdr(const mifx_image2d* im, const char* what, Img& out, bool& packed);
// auto exposure (autoexposure.hip)
mifx_status launch_autoexposure(hipStream_t s, Img color, Img lowRes, float* average, float elapsedTime, int lightAdaptation, bool packedIn = false);
// the same in two steps for row-band sharding: rows [rowBegin, rowEnd) of the 64x64 low-resolution luminance, then the reduction + adaptation from the stored plane
mifx_status launch_autoexposure_rows(hipStream_t s, Img color, Img lowRes, int rowBegin, int rowEnd, bool packedIn = false);
mifx_status launch_autoexposure_reduce(hipStream_t s, Img lowRes, float* average, float elapsedTime, int lightAdaptation);
mifx_status launch_postfx_prep(hipStream_t s, Img depth, Img motion, Img reproj, Img closest, const CamK& cur, const CamK& prev, const uint8_t* sobol, const uint8_t* tile, Img noiseXY,
                               Img noiseZW, uint32_t frame, bool halfPrecisionDepth = false);
mifx_status launch_depth16_copy(hipStream_t s, Img in, Img out); // out = what an R16_UNORM copy of `in` keeps (native-storage build; identity in the fp32 build)
// SSAO (ssao.hip)
mifx_status launch_ssao_prefilter_pyramid(hipStream_t s, const Pyr& p, const Pyr& camz, const CamK& cam, const mifx_ssao_attribs& a, bool depth16 = false);
mifx_status launch_ssao_compute_ao(hipStream_t s, const Pyr& depthPyr, const Pyr& camzPyr, Img normal, Img noiseZW, Img out, const CamK& cam, const mifx_ssao_attribs& a,
                                   bool halfResolution, bool halfPrecisionDepth);
mifx_status launch_ssao_downsample_depth(hipStream_t s, Img depth, Img out);
mifx_status launch_ssao_depth_to_camz(hipStream_t s, Img depth, Img camz, const CamK& cam);
mifx_status launch_ssao_bilateral_upsample(hipStream_t s, Img depth, Img occlusion, Img out, const CamK& cam);
// the fused resolve (ssao.hip): what the temporal pass writes beside its own targets, and the work lists of the two passes that follow
struct SsaoResolve
{
    Img   depth, resampled, out, out2;
    void* lists;
};
// bytes of `lists` for a w x h frame: per 64x4 workgroup of the temporal pass two counters and two segments of 256 entries
inline size_t ssao_resolve_list_bytes(uint32_t w, uint32_t h) { return size_t((w + 63u) / 64u) * size_t((h + 3u) / 4u) * (2u + 2u * 256u) * 4u; }
mifx_status launch_ssao_temporal(hipStream_t s, Img currAO, Img prevAO, Img prevLen, Img reprojDepth, Img prevDepth, Img motion, Img outAO, Img outLen, const CamK& cur,
                                 const CamK& prev, const mifx_ssao_attribs& a, const SsaoResolve* resolve = nullptr);
mifx_status launch_ssao_resolve_lists(hipStream_t s, const Pyr& aoPyr, const Pyr& depthPyr, Img histLen, Img camz, Img normal, Img rows5, const SsaoResolve& r, const CamK& cam,
                                      const mifx_ssao_attribs& a);
mifx_status launch_ssao_convolute_pyramids(hipStream_t s, const Pyr& ao, const Pyr& depth, bool depth16 = false);
mifx_status launch_ssao_resample(hipStream_t s, const Pyr& aoPyr, const Pyr& depthPyr, Img histLen, Img normal, Img out, const CamK& cam);
mifx_status launch_ssao_spatial(hipStream_t s, Img occl, Img histLen, Img depth, Img camz, Img normal, Img out, Img historyOut, const CamK& cam, const mifx_ssao_attribs& a);
// PBR shade + composite (pbr.hip)
// Optional by-product of the shade (the chain): the roughness / reflection-mask planes of ScreenSpaceReflection's pass R2, whose inputs are the material and
// depth texels the shade reads anyway.  `enabled == 0`: nothing is written.
struct SsrMaskOut
{
    Img      roughness, mask;
    float    threshold;
    int      perceptual;
    unsigned channel;
    int      enabled;
};
mifx_status launch_pbr_shade(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl,
                             const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_spec, int row_begin, int row_end, bool reversedDepth,
                             const mifx_pbr_shadows* shadows = nullptr, const SsrMaskOut* ssrMask = nullptr);
// the sharded SSR's hit fetch with the layered body: `out_radiance` of launch_pbr_shade_layers is then the plane phase 0 shaded rows [shadedBegin, shadedEnd) of
struct LayeredHitFetch
{
    Img rays, coords;
    int shadedBegin, shadedEnd;
};
mifx_status launch_pbr_shade_layers(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_pbr_layers& layers, const mifx_camera_attribs& camera,
                                    const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl, const float background[4], const mifx_image2d* out_radiance, const mifx_image2d* out_spec,
                                    int row_begin, int row_end, bool reversedDepth, const mifx_pbr_shadows* shadows = nullptr, const LayeredHitFetch* hit = nullptr);
// Row-band sharding: the colour of every ray hit that ssr_intersection_kernel recorded in `hitCoords` (packed x | y << 16; 0xffffffff = outside the frame) goes into
// xyz of `rays` (w = the confidence the march wrote): loaded from `radiance` when the hit row lies in [shadedBegin, shadedEnd) -- the rows this rank shaded -- and
// otherwise computed on the spot by the shade kernel's own body for that one pixel (the G-buffer and the IBL maps are whole on every rank): no radiance exchange.
mifx_status launch_pbr_hit_fetch(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a, const mifx_ibl* ibl,
                                 const float background[4], Img rays, Img hitCoords, const mifx_image2d* radiance, int shadedBegin, int shadedEnd, bool reversedDepth);
mifx_status launch_pbr_shade_native(hipStream_t s, IblApronCache& iblApron, const mifx_gbuffer_native* g, const mifx_camera_attribs& camera, const mifx_pbr_shade_attribs& a,
                                    const mifx_ibl* ibl, const float background[4], const mifx_native_image* out_radiance, const mifx_native_image* out_spec, bool reversedDepth);
// r7 != nullptr: the chain's composite evaluates SSR's bilateral cleanup itself (a.ssr is then not read)
mifx_status launch_composite(hipStream_t s, const mifx_composite_attribs& a, const mifx_image2d* out, int row_begin, int row_end, const SsrCleanupIn* r7 = nullptr);
mifx_status launch_specgloss_material(hipStream_t s, Img baseColor, Img physicalDesc, Img out);
// Bloom (bloom.hip) + TAA (taa.hip)
// (packedInput: `in` / `input` is depth of field's R11G11B10_FLOAT output plane -- native-storage build; see to_img_hdr)
mifx_status launch_bloom_prefilter(hipStream_t s, Img in, Img out, const mifx_bloom_attribs& a, bool packedInput = false);
mifx_status launch_bloom_downsample(hipStream_t s, Img in, Img out);
mifx_status launch_bloom_upsample(hipStream_t s, Img input, Img down, Img out, const mifx_bloom_attribs& a, bool final_pass, bool packedInput = false);
bool        bloom_tail_fits(const Img* down, int count);
mifx_status launch_bloom_tail(hipStream_t s, const Img* down, const Img* up, int count); // the small levels of the pyramid, down and up, in one workgroup
mifx_status launch_bloom_final_tonemap(hipStream_t s, Img input, Img down, Img out, Img ldr, const mifx_bloom_attribs& a, const mifx_tone_mapping_attribs& attr, float ave_log_lum,
                                       uint32_t flags, bool writeBloomOutput = true, bool packedInput = false); // the final up-sample + the chain's copy-frame ToneMap in one pass
// fused != nullptr: the colour TAA accumulates is the chain's composite, evaluated inside the kernel (taa.hip); currColor is then not read
struct TaaFusedComposite
{
    const mifx_composite_attribs* attribs; // as for launch_composite (its `ssr` plane is not read: the cleanup is evaluated in place)
    const SsrCleanupIn*           r7;
};
mifx_status launch_taa(hipStream_t s, Img currColor, Img prevColor, Img motion, Img reprojDepth, Img prevDepth, Img out, const CamK& cur, const CamK& prev,
                       const mifx_taa_attribs& a, uint32_t flags, const TaaFusedComposite* fused = nullptr);
// Depth of field (dof.hip)
mifx_status launch_dof_coc(hipStream_t s, Img depth, Img out, const mifx_camera_attribs& cam, float maxCoC);
mifx_status launch_dof_temporal_coc(hipStream_t s, Img curr, Img prev, Img motion, Img out, const mifx_camera_attribs& cam, float stability);
mifx_status launch_dof_dilation(hipStream_t s, Img coc, const Img levels[3]);
mifx_status launch_dof_blur(hipStream_t s, Img in, Img out, const float weights[13]);
mifx_status launch_dof_prefilter(hipStream_t s, Img color, Img coc, Img dilation, Img outNear, Img outFar);
mifx_status launch_dof_bokeh_gather(hipStream_t s, Img nearTex, Img farTex, Img radiance, Img outNear, Img outFar, const float* kernel, int sampleCount, float maxCoC, float aspect,
                                    bool karis);
mifx_status launch_dof_bokeh_fill(hipStream_t s, Img nearTex, Img farTex, Img outNear, Img outFar, const float* kernel, int sampleCount, float maxCoC, float aspect);
mifx_status launch_dof_postfilter(hipStream_t s, Img nearTex, Img farTex, Img outNear, Img outFar);
mifx_status launch_dof_combine(hipStream_t s, Img color, Img nearTex, Img farTex, Img out, float alpha);
// native formats (formats.hip)
uint32_t    native_texel_size(uint32_t fmt);
mifx_status launch_image_import(hipStream_t s, const mifx_native_image* src, const mifx_image2d* dst);
mifx_status launch_image_export(hipStream_t s, const mifx_image2d* src, const mifx_native_image* dst);
// SSR (ssr.hip)
mifx_status launch_ssr_hiz_pyramid(hipStream_t s, const Pyr& p, Img level0Copy, bool reversedDepth);
mifx_status launch_ssr_mask_roughness(hipStream_t s, Img material, Img depth, Img roughness, Img mask, const mifx_ssr_attribs& a, bool reversedDepth);
mifx_status launch_ssr_intersection(hipStream_t s, Img radiance, Img normal, Img roughness, Img noiseXY, const HizSlab& hiz, Img mask, Img motion, Img outSpec, Img outDirPdf,
                                    const CamK& cam, const mifx_ssr_attribs& a, bool previousFrame, bool halfResolution, Img hitCoords = Img{}, int localBegin = 0, int localEnd = 0);
// (hitCoords: row-band sharding -- hits outside the rows [localBegin, localEnd) of `radiance` are recorded for launch_pbr_hit_fetch instead of loaded)
mifx_status launch_ssr_downsampled_mask(hipStream_t s, Img roughness, Img depth, Img mask, const mifx_ssr_attribs& a, bool reversedDepth);
mifx_status launch_ssr_spatial(hipStream_t s, Img roughness, Img normal, Img depth, Img dirPdf, Img spec, Img mask, Img outRad, Img outVar, Img outDepth, const CamK& cam,
                               const mifx_ssr_attribs& a, bool halfResolution);
mifx_status launch_ssr_temporal(hipStream_t s, Img motion, Img hitDepth, Img reprojDepth, Img currRad, Img currVar, Img prevDepth, Img prevRad, Img prevVar, Img mask, Img outRad,
                                Img outVar, const CamK& cur, const CamK& prev, const mifx_ssr_attribs& a);
mifx_status launch_ssr_bilateral(hipStream_t s, Img normal, const SsrCleanupIn& in, Img out, const CamK& cam);
// IBL precompute (ibl.hip)
mifx_status launch_ibl_brdf_lut(hipStream_t s, Img out, uint32_t num_samples);
mifx_status launch_ibl_prefilter(hipStream_t s, const mifx_cubemap* env, const mifx_spheremap* sphere, void* out, uint32_t out_size, float roughness, uint32_t num_samples);
mifx_status launch_ibl_irradiance(hipStream_t s, const mifx_cubemap* env, const mifx_spheremap* sphere, void* out, uint32_t out_size, uint32_t num_samples);
mifx_status launch_envmap(hipStream_t s, const mifx_envmap_render_attribs& a, const mifx_tone_mapping_attribs& tm, const mifx_camera_attribs& cam, const mifx_camera_attribs& prev,
                          Img depth, Img color, Img motion);

} // namespace mifx
