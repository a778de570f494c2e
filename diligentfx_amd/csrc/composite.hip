// composite.hip -- (M1) the SSR / SSAO composite of the chain, Hydrogent/shaders/HnPostProcess.psh:145-185.  116 B/px.
//
// A translation unit of its own since round 3: the chain's instance of the kernel evaluates ScreenSpaceReflection's bilateral cleanup (R7) in place
// (mifx_ssr_cleanup.h), whose arithmetic must not be contracted, while the composite's own smooth BRDF arithmetic keeps its fused multiply-adds -- build.py compiles
// the FMA sources with -ffp-contract=fast-honor-pragmas, under which the `#pragma clang fp contract(off)` of the cleanup holds (plain `fast` lets the backend fuse
// across it).
#include "mifx_host.h"
#include "mifx_pbr.h"
#include "mifx_effects.h"
#include "mifx_tonemap.h"
#include "mifx_ssr_cleanup.h"
#include "mifx_composite.h"

namespace mifx
{
mifx_status make_lutk(const mifx_image2d* im, LutK& k); // pbr.hip

// ------------------------------------------------------------------------------------------------ M1 composite
// (the per-pixel body: mifx_composite.h)
template <int TM_MODE, bool FUSE_R7>
__global__ __launch_bounds__(256) void composite_kernel(Img color, Img specIBL, Img ssr, Img ssao, Img normalTex, Img baseColor, Img material, LutK lut, Img out, CamK cam,
                                                        float ssrScaleAttr, float ssaoScaleAttr, ToneMapK tm, SsrCleanupIn r7)
{
    int x, y;
    if (!pixel_xy_dir<2>(out, x, y)) return;
    v4 result;
    composite_pixel<TM_MODE, FUSE_R7>(result, x, y, color, specIBL, ssr, ssao, normalTex, baseColor, material, lut, out.w, out.h, cam, ssrScaleAttr, ssaoScaleAttr, tm, r7);
    st<v4>(out, x, y, result);
}

mifx_status launch_composite(hipStream_t s, const mifx_composite_attribs& a, const mifx_image2d* out_img, int row_begin, int row_end, const SsrCleanupIn* r7)
{
    Img color, sibl, ssr, ssao, nrm, bc, mat, out;
    MIFX_CHECK(to_img(out_img, MIFX_FORMAT_F32X4, "out", out));
    out = rows_of(out, row_begin, row_end);
    const uint32_t W = out_img->width, H = out_img->height;
    MIFX_CHECK(to_img_wh(a.color, MIFX_FORMAT_F32X4, W, H, "color", color));
    MIFX_CHECK(to_img_wh(a.specular_ibl, MIFX_FORMAT_F32X4, W, H, "specular_ibl", sibl));
    if (r7) ssr = Img{};
    else MIFX_CHECK(to_img_wh(a.ssr, MIFX_FORMAT_F32X4, W, H, "ssr", ssr));
    MIFX_CHECK(to_img_wh(a.ssao, MIFX_PLANE_AO, W, H, "ssao", ssao));
    MIFX_CHECK(to_img_wh(a.normal, MIFX_FORMAT_F32X4, W, H, "normal", nrm));
    MIFX_CHECK(to_img_wh(a.base_color, MIFX_FORMAT_F32X4, W, H, "base_color", bc));
    MIFX_CHECK(to_img_wh(a.material, MIFX_FORMAT_F32X4, W, H, "material", mat));
    MIFX_REQUIRE(a.camera != nullptr, "camera must not be null");
    LutK lut;
    MIFX_CHECK(make_lutk(a.brdf_lut, lut));
    int mode = a.tone_mapping ? a.tone_mapping->iToneMappingMode : 0;
    MIFX_REQUIRE(mode >= 0 && mode <= MIFX_TONE_MAPPING_MODE_COMMERCE, "unknown tone mapping mode %d", mode);
    // HnPostProcess.psh:183-185: ToneMap(Color, attribs, AverageLogLum * exp2(-fExposure))
    const ToneMapK tm = a.tone_mapping ? make_tonemapk(*a.tone_mapping, a.ave_log_lum * m_exp2(-a.camera->fExposure)) : ToneMapK{};
    const CamK cam = make_camk(*a.camera);
    const dim3 block(64, 4, 1), grid = grid2d(out, block);
    if (r7)
    {
        // (the chain composites without a tone map -- TAA follows; the fused instance exists for that mode only)
        MIFX_REQUIRE(mode == MIFX_TONE_MAPPING_MODE_NONE, "composite with the fused SSR cleanup: tone mapping mode %d not instantiated", mode);
        hipLaunchKernelGGL((composite_kernel<MIFX_TONE_MAPPING_MODE_NONE, true>), grid, block, 0, s, color, sibl, ssr, ssao, nrm, bc, mat, lut, out, cam, a.ssr_scale, a.ssao_scale, tm, *r7);
    }
    else
    {
#define MIFX_COMP(M) hipLaunchKernelGGL((composite_kernel<M, false>), grid, block, 0, s, color, sibl, ssr, ssao, nrm, bc, mat, lut, out, cam, a.ssr_scale, a.ssao_scale, tm, SsrCleanupIn{})
        MIFX_TONEMAP_DISPATCH(mode, MIFX_COMP)
#undef MIFX_COMP
    }
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
