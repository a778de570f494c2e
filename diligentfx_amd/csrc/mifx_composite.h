// mifx_composite.h -- (M1) one pixel of the SSR / SSAO composite of the chain, Hydrogent/shaders/HnPostProcess.psh:145-185: the body of composite_kernel (composite.hip),
// in a header so that the test suite can also compile it for the host (tests/host_kernels/chain_host.cpp).
#pragma once
#include "mifx_pbr.h"
#include "mifx_effects.h"
#include "mifx_tonemap.h"
#include "mifx_ssr_cleanup.h"

namespace mifx
{
// FUSE_R7: the reflection is SSR's bilateral cleanup (pass R7) evaluated here for this pixel from the effect's accumulated radiance / variance instead of a load of
// the plane R7 would have written -- this kernel is that plane's only consumer in the chain (mifx_ssr_cleanup.h; `ssr` is not read).  outW / outH: the size of the target.
template <int TM_MODE, bool FUSE_R7>
MIFX_D void composite_pixel(v4& result, int x, int y, const Img& color, const Img& specIBL, const Img& ssr, const Img& ssao, const Img& normalTex, const Img& baseColor, const Img& material,
                          const LutK& lut, int outW, int outH, const CamK& cam, float ssrScaleAttr, float ssaoScaleAttr, const ToneMapK& tm, const SsrCleanupIn& r7)
{
    // (loads grouped by what they depend on: the colour -- whose alpha decides whether anything else is read -- with the reflection mask; then every other plane of
    //  the pixel at once, the inputs of the fused cleanup included; then the LUT taps, which need the roughness and the normal)
    v4 c = ld_once<v4>(color, x, y);
    const float maskValue = FUSE_R7 ? ld<mask_t>(r7.mask, x, y) : 1.0f;
    const float opacity  = c.w;
    const float ssrScale = ssrScaleAttr * opacity;
    const float ssaoScale = ssaoScaleAttr * opacity;
    const float ao = ssaoScale > 0.0f ? ld_once<ao_t>(ssao, x, y) : 1.0f;
    v3 rgb = xyz(c);
    if (ssrScale > 0.0f)
    {
        const v4 sibl = ld_once<v4>(specIBL, x, y);
        const v3 N    = xyz(ld<v4>(normalTex, x, y));
        const v4 bc   = ld_once<v4>(baseColor, x, y);
        const v4 mat  = ld_once<v4>(material, x, y);
        // (quantize_v4: what the store into the pass's 4-channel target and the load back from it do to the value -- nothing in the fp32 build, a binary16 rounding in
        //  the native-storage build, where the fused and the separate pass must still agree)
        const v4 refl = FUSE_R7 ? quantize_v4(ssr_bilateral_cleanup(x, y, N, maskValue, normalTex, r7, cam.proj, int(cam.vw), int(cam.vh))) : ld<v4>(ssr, x, y);
        const SurfaceReflectance srf = surface_reflectance_mr(xyz(bc), saturate(mat.y), saturate(mat.x));
        // f2NormalizedXY of the pixel centre, depth 0.5 => a point on the view ray
        const v2 ndc{fdiv(2.0f * (float(x) + 0.5f), float(outW)) - 1.0f, 1.0f - fdiv(2.0f * (float(y) + 0.5f), float(outH))};
        const v4 wp   = mul(v4{ndc.x, ndc.y, 0.5f, 1.0f}, cam.viewProjInv);
        const v3 view = normalize(v3{cam.pos[0], cam.pos[1], cam.pos[2]} - xyz(wp) / wp.w);
        const IBLInfo ibl = ibl_sampling_info(srf, lut, N, view);
        const v3 s = specular_ibl_ggx(ibl, xyz(refl));
        rgb = rgb + (s - xyz(sibl)) * refl.w * ssrScale;
    }
    if (ssaoScale > 0.0f) rgb = rgb * lerpf(1.0f, ao, ssaoScale);
    if (TM_MODE != MIFX_TONE_MAPPING_MODE_NONE) rgb = tone_map<TM_MODE>(rgb, tm);
    result = mk4(rgb, c.w);
}
} // namespace mifx
