// ssr_temporal.hip -- ScreenSpaceReflection passes R6 (temporal accumulation) and R7 (bilateral cleanup); R1-R5 are in ssr.hip, R4 in ssr_trace.hip.
// Math follows Shaders/PostProcess/ScreenSpaceReflection/private/SSR_ComputeTemporalAccumulation.fx and SSR_ComputeBilateralCleanup.fx.
// A file of its own so that its floating-point contraction policy can differ from R5's (diligentfx_amd/build.py: FMA_SOURCES).
#include "mifx_host.h"
#include "mifx_effects.h"
#include "mifx_pbr.h"
#include "mifx_ssr_cleanup.h"
#include "mifx_ssr_temporal.h"

namespace mifx
{
// ------------------------------------------------------------------------------------------------ R6: temporal accumulation (SSR_ComputeTemporalAccumulation.fx:104-275)
// (re-measured over 60 frames in round 3: 6 waves 127.0 us, 5 waves 122.1 us, 8 waves 209.9 us (spills); profiles/r03_ab_occupancy_hints.txt)
#ifndef MIFX_R6_WAVES
#define MIFX_R6_WAVES 5
#endif
__global__ __launch_bounds__(256) MIFX_WAVES(MIFX_R6_WAVES) void ssr_temporal_kernel(Img motionTex, Img hitDepthTex, Img currDepth /*reprojected*/, Img currRad, Img currVar, Img prevDepth, Img prevRad,
                                                           Img prevVar, Img mask, Img outRad, Img outVar, CamK cur, CamK prev, SsrK k)
{
    int x, y;
    if (!pixel_xy(outRad, x, y)) return;
    ssr_temporal_pixel(x, y, motionTex, hitDepthTex, currDepth, currRad, currVar, prevDepth, prevRad, prevVar, mask, outRad, outVar, cur, prev, k);
}

// ------------------------------------------------------------------------------------------------ R7: bilateral cleanup (SSR_ComputeBilateralCleanup.fx:49-103; body in mifx_ssr_cleanup.h)
__global__ __launch_bounds__(256) void ssr_bilateral_kernel(Img normalTex, SsrCleanupIn in, Img out, CamK cam)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    const float m = ld<mask_t>(in.mask, x, y);
    st<v4>(out, x, y, ssr_bilateral_cleanup(x, y, xyz(ld<v4>(normalTex, x, y)), m, normalTex, in, cam.proj, int(cam.vw), int(cam.vh)));
}

static const dim3 kBlock(64, 4, 1);
#define MIFX_LAUNCH_END()              \
    MIFX_HIP_CHECK(hipGetLastError()); \
    return MIFX_OK

mifx_status launch_ssr_temporal(hipStream_t s, Img motion, Img hitDepth, Img reprojDepth, Img currRad, Img currVar, Img prevDepth, Img prevRad, Img prevVar, Img mask, Img outRad,
                                Img outVar, const CamK& cur, const CamK& prev, const mifx_ssr_attribs& a)
{
        hipLaunchKernelGGL(ssr_temporal_kernel, grid2d(outRad, kBlock), kBlock, 0, s, motion, hitDepth, reprojDepth, currRad, currVar, prevDepth, prevRad, prevVar, mask,
                       outRad, outVar, cur, prev, make_k(a, cur.reversedDepth != 0));
    MIFX_LAUNCH_END();
}
mifx_status launch_ssr_bilateral(hipStream_t s, Img normal, const SsrCleanupIn& in, Img out, const CamK& cam)
{
    hipLaunchKernelGGL(ssr_bilateral_kernel, grid2d(out, kBlock), kBlock, 0, s, normal, in, out, cam);
    MIFX_LAUNCH_END();
}
} // namespace mifx
