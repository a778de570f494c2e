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

namespace mifx
{
mifx_status make_lutk(const mifx_image2d* im, LutK& k); // pbr.hip

// ------------------------------------------------------------------------------------------------ M1 composite
// FUSE_R7: the reflection is SSR's bilateral cleanup (pass R7) evaluated here for this pixel from the effect's accumulated radiance / variance instead of a load of
// the plane R7 would have written -- this kernel is that plane's only consumer in the chain (mifx_ssr_cleanup.h; `ssr` is not read).
template <int TM_MODE, bool FUSE_R7>
__global__ __launch_bounds__(256) void composite_kernel(Img color, Img specIBL, Img ssr, Img ssao, Img normalTex, Img baseColor, Img material, LutK lut, Img out, CamK cam,
                                                        float ssrScaleAttr, float ssaoScaleAttr, ToneMapK tm, SsrCleanupIn r7)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    // (loads grouped by what they depend on: the colour -- whose alpha decides whether anything else is read -- with the reflection mask; then every other plane of
    //  the pixel at once, the inputs of the fused cleanup included; then the LUT taps, which need the roughness and the normal)
    v4 c = ld<v4>(color, x, y);
    const float maskValue = FUSE_R7 ? ld<mask_t>(r7.mask, x, y) : 1.0f;
    const float opacity  = c.w;
    const float ssrScale = ssrScaleAttr * opacity;
    const float ssaoScale = ssaoScaleAttr * opacity;
    const float ao = ssaoScale > 0.0f ? ld<ao_t>(ssao, x, y) : 1.0f;
    v3 rgb = xyz(c);
    if (ssrScale > 0.0f)
    {
        const v4 sibl = ld<v4>(specIBL, x, y);
        const v3 N    = xyz(ld<v4>(normalTex, x, y));
        const v4 bc   = ld<v4>(baseColor, x, y);
        const v4 mat  = ld<v4>(material, x, y);
        // (quantize_v4: what the store into the pass's 4-channel target and the load back from it do to the value -- nothing in the fp32 build, a binary16 rounding in
        //  the native-storage build, where the fused and the separate pass must still agree)
        const v4 refl = FUSE_R7 ? quantize_v4(ssr_bilateral_cleanup(x, y, N, maskValue, normalTex, r7, cam.proj, int(cam.vw), int(cam.vh))) : ld<v4>(ssr, x, y);
        const SurfaceReflectance srf = surface_reflectance_mr(xyz(bc), saturate(mat.y), saturate(mat.x));
        // f2NormalizedXY of the pixel centre, depth 0.5 => a point on the view ray
        const v2 ndc{fdiv(2.0f * (float(x) + 0.5f), float(out.w)) - 1.0f, 1.0f - fdiv(2.0f * (float(y) + 0.5f), float(out.h))};
        const v4 wp   = mul(v4{ndc.x, ndc.y, 0.5f, 1.0f}, cam.viewProjInv);
        const v3 view = normalize(v3{cam.pos[0], cam.pos[1], cam.pos[2]} - xyz(wp) / wp.w);
        const IBLInfo ibl = ibl_sampling_info(srf, lut, N, view);
        const v3 s = specular_ibl_ggx(ibl, xyz(refl));
        rgb = rgb + (s - xyz(sibl)) * refl.w * ssrScale;
    }
    if (ssaoScale > 0.0f) rgb = rgb * lerpf(1.0f, ao, ssaoScale);
    if (TM_MODE != MIFX_TONE_MAPPING_MODE_NONE) rgb = tone_map<TM_MODE>(rgb, tm);
    st<v4>(out, x, y, mk4(rgb, c.w));
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
