// TEST INFRASTRUCTURE ONLY.  Two header-only pieces of the timed chain compiled for the HOST (see layers_host.cpp for the method and for what it can and cannot show):
//   * ToneMap() of every TONE_MAPPING_MODE (diligentfx_amd/csrc/mifx_tonemap.h: tone_map<MODE>, linear_to_srgb) -- the body of tonemap_kernel and of the tail that Bloom's
//     final up-sample and the composite fuse;
//   * SSR's pass R7, the bilateral cleanup (mifx_ssr_cleanup.h: ssr_bilateral_cleanup) -- the body of ssr_bilateral_kernel and of the composite kernel's fused variant;
//   * SSR's pass R6, the temporal accumulation (mifx_ssr_temporal.h: ssr_temporal_pixel) -- the body of ssr_temporal_kernel;
//   * M1, the SSR / SSAO composite (mifx_composite.h: composite_pixel) -- the body of composite_kernel, with the reflection read from R7's plane or with R7 evaluated in place.
// Nothing in diligentfx_amd/ builds, loads or calls this.
#include <hip/hip_runtime.h>
#undef __device__
#define __device__ __attribute__((host)) __attribute__((device))
#include "mifx_tonemap.h"
#include "mifx_ssr_cleanup.h"
#include "mifx_composite.h"
#include "mifx_ssr_temporal.h"
#include <cstring>

using namespace mifx;

static CamK host_camk(const mifx_camera_attribs* c) // make_camk (mifx_core.cpp)
{
    CamK k{};
    std::memcpy(k.view.m, c->mView, 64); std::memcpy(k.proj.m, c->mProj, 64); std::memcpy(k.viewProj.m, c->mViewProj, 64);
    std::memcpy(k.viewInv.m, c->mViewInv, 64); std::memcpy(k.viewProjInv.m, c->mViewProjInv, 64);
    for (int i = 0; i < 3; ++i) k.pos[i] = c->f4Position[i];
    k.vw = c->f4ViewportSize[0]; k.vh = c->f4ViewportSize[1]; k.ivw = c->f4ViewportSize[2]; k.ivh = c->f4ViewportSize[3];
    k.jx = c->f2Jitter[0]; k.jy = c->f2Jitter[1];
    k.frameIndex = c->uiFrameIndex;
    return k;
}

extern "C" {
// in / out: w x h float4 texels, tightly packed
int mifx_host_tonemap(const float* in, float* out, int w, int h, const mifx_tone_mapping_attribs* attribs, float ave_log_lum, int srgb)
{
    const ToneMapK k = make_tonemapk(*attribs, ave_log_lum);
    const int mode = attribs->iToneMappingMode;
    if (mode < 0 || mode > 11) return 1;
#define RUN(M)                                                                      \
    for (long i = 0; i < long(w) * h; ++i)                                          \
    {                                                                               \
        const v4 c{in[4 * i], in[4 * i + 1], in[4 * i + 2], in[4 * i + 3]};         \
        v3 t = tone_map<M>(xyz(c), k);                                              \
        if (srgb) t = linear_to_srgb(t);                                            \
        out[4 * i] = t.x; out[4 * i + 1] = t.y; out[4 * i + 2] = t.z; out[4 * i + 3] = c.w; \
    }
    MIFX_TONEMAP_DISPATCH(mode, RUN)
#undef RUN
    return 0;
}

// depth, roughness, variance, mask: w x h floats; normal, radiance, out: w x h float4; proj: CameraAttribs::mProj (16 floats)
int mifx_host_ssr_bilateral_cleanup(const float* depth, const float* normal, const float* roughness, const float* radiance, const float* variance, const float* mask, float* out, int w,
                                    int h, const float* proj, float roughness_threshold, float spatial_sigma_factor, float alpha_interpolation, int reversed_depth)
{
    auto img = [&](const float* p, int c) { return Img{reinterpret_cast<unsigned char*>(const_cast<float*>(p)), w, h, w * c * 4, 0, 0}; };
    SsrCleanupIn in{img(depth, 1), img(roughness, 1), img(radiance, 4), img(variance, 1), img(mask, 1), roughness_threshold, spatial_sigma_factor, alpha_interpolation, reversed_depth};
    const Img normalTex = img(normal, 4);
    m44 P;
    std::memcpy(P.m, proj, 64);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            const v4 n = ld<v4>(normalTex, x, y);
            const v4 r = ssr_bilateral_cleanup(x, y, xyz(n), ld<mask_t>(in.mask, x, y), normalTex, in, P, w, h);
            float* o = out + 4 * (size_t(y) * w + x);
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
        }
    return 0;
}
// color, specular_ibl, ssr (or null: R7 evaluated in place from radiance / variance / mask / depth / roughness), normal, base_color, material, out: w x h float4; ssao: w x h floats;
// lut: lw x lh texels of lc floats; camera: CameraAttribs
int mifx_host_composite(const float* color, const float* specular_ibl, const float* ssr, const float* ssao, const float* normal, const float* base_color, const float* material,
                        const float* lut, int lw, int lh, int lc, float* out, int w, int h, const mifx_camera_attribs* camera, float ssr_scale, float ssao_scale, const float* depth,
                        const float* roughness, const float* radiance, const float* variance, const float* mask, float roughness_threshold, float spatial_sigma_factor,
                        float alpha_interpolation)
{
    auto img = [&](const float* p, int c) { return p ? Img{reinterpret_cast<unsigned char*>(const_cast<float*>(p)), w, h, w * c * 4, 0, 0} : Img{}; };
    const Img c0 = img(color, 4), sibl = img(specular_ibl, 4), refl = img(ssr, 4), ao = img(ssao, 1), nrm = img(normal, 4), bc = img(base_color, 4), mat = img(material, 4);
    const LutK lutk{lut, lw, lh, lw * lc, lc};
    CamK cam{}; // make_camk (mifx_core.cpp)
    std::memcpy(cam.view.m, camera->mView, 64); std::memcpy(cam.proj.m, camera->mProj, 64); std::memcpy(cam.viewProj.m, camera->mViewProj, 64);
    std::memcpy(cam.viewInv.m, camera->mViewInv, 64); std::memcpy(cam.viewProjInv.m, camera->mViewProjInv, 64);
    for (int i = 0; i < 3; ++i) cam.pos[i] = camera->f4Position[i];
    cam.vw = camera->f4ViewportSize[0]; cam.vh = camera->f4ViewportSize[1]; cam.ivw = camera->f4ViewportSize[2]; cam.ivh = camera->f4ViewportSize[3];
    const SsrCleanupIn r7{img(depth, 1), img(roughness, 1), img(radiance, 4), img(variance, 1), img(mask, 1), roughness_threshold, spatial_sigma_factor, alpha_interpolation, 0};
    const ToneMapK tm{};
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x)
        {
            v4 r;
            if (ssr != nullptr) composite_pixel<MIFX_TONE_MAPPING_MODE_NONE, false>(r, x, y, c0, sibl, refl, ao, nrm, bc, mat, lutk, w, h, cam, ssr_scale, ssao_scale, tm, r7);
            else composite_pixel<MIFX_TONE_MAPPING_MODE_NONE, true>(r, x, y, c0, sibl, refl, ao, nrm, bc, mat, lutk, w, h, cam, ssr_scale, ssao_scale, tm, r7);
            float* o = out + 4 * (size_t(y) * w + x);
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = r.w;
        }
    return 0;
}
// R6: motion (c = 2), hit depth / reprojected depth / previous depth / current and previous variance / mask (c = 1), current and previous radiance (c = 4);
// out_radiance / out_variance: the history slot of this frame, holding what the frame before last left (the pass writes under the mask only)
int mifx_host_ssr_temporal(const float* motion, const float* hit_depth, const float* reprojected_depth, const float* curr_radiance, const float* curr_variance,
                           const float* prev_depth, const float* prev_radiance, const float* prev_variance, const float* mask, float* out_radiance, float* out_variance, int w, int h,
                           const mifx_camera_attribs* camera, const mifx_camera_attribs* prev_camera, const mifx_ssr_attribs* attribs)
{
    auto img = [&](const float* p, int c) { return Img{reinterpret_cast<unsigned char*>(const_cast<float*>(p)), w, h, w * c * 4, 0, 0}; };
    const Img mo = img(motion, 2), hd = img(hit_depth, 1), rd = img(reprojected_depth, 1), cr = img(curr_radiance, 4), cv = img(curr_variance, 1), pd = img(prev_depth, 1),
              pr = img(prev_radiance, 4), pv = img(prev_variance, 1), mk = img(mask, 1), orad = img(out_radiance, 4), ovar = img(out_variance, 1);
    const CamK cur = host_camk(camera), prev = host_camk(prev_camera);
    const SsrK k = make_k(*attribs, false);
#pragma omp parallel for schedule(dynamic, 4)
    for (int y = 0; y < h; ++y)
        for (int x = 0; x < w; ++x) ssr_temporal_pixel(x, y, mo, hd, rd, cr, cv, pd, pr, pv, mk, orad, ovar, cur, prev, k);
    return 0;
}
}
