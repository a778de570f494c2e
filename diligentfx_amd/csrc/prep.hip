// prep.hip -- PostFXContext::Execute passes (PostProcess/Common/src/PostFXContext.cpp:287-338):
//   C1 blue noise      Shaders/Common/private/ComputeBlueNoiseTexture.fx:18-89   (2 x 128x128 float2, UNORM8-quantised)
//   C2 reprojected depth  .../ComputeReprojectedDepth.fx:18-30  } fused into one kernel: both are functions of the
//   C3 closest motion     .../ComputeClosestMotion.fx:24-55     } current depth tile (20 B/px read, 12 B/px written)
//   C4 previous depth copy (PostFXContext.cpp:657-676) is an alias of the borrowed plane, no kernel.
#include "mifx_host.h"

namespace mifx
{
// ------------------------------------------------------------------------------------------------ C1
// C1 must be BIT-exact (the noise is quantised to UNORM8 and steers every stochastic pass): no FMA contraction, correctly rounded divisions.
#pragma clang fp contract(off)
__device__ __forceinline__ float blue_noise_sample(const uint8_t* sobol, const uint8_t* tile, uint32_t px, uint32_t py, uint32_t dim)
{
    px &= 127u; py &= 127u; dim &= 255u;
    uint32_t value = sobol[dim];
    uint32_t idx   = (dim % 8u) + (px + py * 128u) * 8u; // == x + 512*y of the 512x256 R8_UINT tile texture
    value ^= tile[idx];
    return (float(value) + 0.5f) * (1.0f / 256.0f); // == / 256 exactly (power of two)
}
__device__ __forceinline__ uint32_t hilbert_index(uint32_t px, uint32_t py) // ComputeBlueNoiseTexture.fx:34-57, HILBERT_LEVEL 7
{
    const uint32_t W = 128u;
    px &= W - 1u; py &= W - 1u;
    uint32_t index = 0u;
    for (uint32_t lvl = W / 2u; lvl > 0u; lvl /= 2u)
    {
        uint32_t rx = (px & lvl) > 0u ? 1u : 0u;
        uint32_t ry = (py & lvl) > 0u ? 1u : 0u;
        index += lvl * lvl * ((3u * rx) ^ ry);
        if (ry == 0u)
        {
            if (rx == 1u) { px = (W - 1u) - px; py = (W - 1u) - py; }
            uint32_t t = px; px = py; py = t;
        }
    }
    return index;
}
__device__ __forceinline__ float unorm8(float v) // RG8_UNORM render-target store + load (PostFXContext.cpp:200)
{
    v = saturate(v);
    // n / 255 must be the correctly rounded fp32 quotient (what a UNORM8 -> float conversion returns); the fp64 quotient rounded to
    // fp32 equals it for all 256 values of n (checked exhaustively) and is immune to the fast fp32 division the library is built with
    return float(double(floorf(v * 255.0f + 0.5f)) / 255.0);
}
struct NoiseK // C1's inputs and targets; sobol == nullptr: no blue noise in this launch
{
    const uint8_t* sobol;
    const uint8_t* tile;
    Img            xy, zw;
    uint32_t       frame;
};
MIFX_D void blue_noise_texel(const NoiseK& n, uint32_t x, uint32_t y)
{
    // SampleRandomVector2D (:60-68): Heitz sampler + R1 shift
    constexpr float invG1 = 1.0f / 1.61803398875f; // rcp(G), folded at compile time (correctly rounded)
    const float alpha = 0.5f + invG1 * float(n.frame & 0xFFu);
    v2 a{fracf(blue_noise_sample(n.sobol, n.tile, x, y, 0u) + alpha), fracf(blue_noise_sample(n.sobol, n.tile, x, y, 1u) + alpha)};
    // SampleRandomVector1D1D (:71-79): Hilbert-indexed R2 sequence
    uint32_t index = hilbert_index(x, y) + n.frame;
    index += 288u * (n.frame & 127u);
    constexpr float G2 = 1.32471795724474602596f;
    constexpr float ax = 1.0f / G2, ay = 1.0f / (G2 * G2);
    v2 b{fracf(0.5f + float(index) * ax), fracf(0.5f + float(index) * ay)};
    st<v2>(n.xy, x, y, v2{unorm8(a.x), unorm8(a.y)});
    st<v2>(n.zw, x, y, v2{unorm8(b.x), unorm8(b.y)});
}
// ------------------------------------------------------------------------------------------------ C2 + C3
// REV = POSTFX_OPTION_INVERTED_DEPTH (ComputeClosestMotion.fx:5-9,36-40): a template parameter -- as a run-time select in the 3x3 search it cost 40 us
// The blue-noise pass C1 (128 x 128 texels, a 5 us launch of its own) rides along: the workgroups behind the last row of the depth's grid write the two noise planes
// (`noiseRow0` = the first such row of workgroups; 64 of them are used).
template <bool REV> __global__ __launch_bounds__(256) void postfx_prep_kernel(Img depth, Img motion, Img reproj, Img closest, CamK cur, CamK prev, NoiseK noise, unsigned noiseRow0, int halfPrecisionDepth)
{
    if (blockIdx.y >= noiseRow0)
    {
        const unsigned b = (blockIdx.y - noiseRow0) * gridDim.x + blockIdx.x; // 64 workgroups of 64 x 4 texels
        if (noise.sobol != nullptr && b < 64u) blue_noise_texel(noise, (b & 1u) * 64u + threadIdx.x, (b >> 1) * 4u + threadIdx.y);
        return;
    }
    int x, y;
    if (!pixel_xy(depth, x, y)) return;

    // the 3x3 neighbourhood of the depth first (nine independent loads; the centre is C2's input), then the arithmetic: two round trips with the motion tap instead of
    // three.  The search is not clamped in the reference; out-of-bounds Load returns 0 (D3D), which this reproduces.
    float nd[9];
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) nd[(dx + 1) * 3 + (dy + 1)] = ld_zero_f_nb(depth, x + dx, y + dy); // (no branch around the load: 57.6 -> 55.5 us)

    // C3: motion vector of the closest-depth texel of the 3x3 neighbourhood
    float closestDepth = REV ? 0.0f : 1.0f; // DepthFarPlane
    int   ox = 0, oy = 0;
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx)
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
        {
            const float v = nd[(dx + 1) * 3 + (dy + 1)];
            if (REV ? v > closestDepth : v < closestDepth) { ox = dx; oy = dy; closestDepth = v; }
        }
    const v2 cm = ld_zero_v2(motion, x + ox, y + oy);

    // C2: unproject with the current inverse view-projection (jitter removed), reproject with the previous one
    const float d = nd[4];
    v3 sc{(float(x) + 0.5f) * cur.ivw, (float(y) + 0.5f) * cur.ivh, d};
    sc.x += 0.5f * cur.jx;
    sc.y += -0.5f * cur.jy;
    const v3 world = inv_project_position(sc, cur.viewProjInv);
    const v3 prevc = project_position(world, prev.viewProj);
    st<float>(reproj, x, y, depth16(prevc.z, halfPrecisionDepth));
    st<cm_t>(closest, x, y, cm);
}
// what an R16_UNORM copy of a depth plane keeps (native-storage build with FEATURE_FLAG_HALF_PRECISION_DEPTH: PostFXContext's previous depth, mip 0 of SSAO's depth pyramids)
__global__ __launch_bounds__(256) void depth16_copy_kernel(Img in, Img out)
{
    int x, y;
    if (!pixel_xy(out, x, y)) return;
    st<float>(out, x, y, depth16(ld<float>(in, x, y), 1));
}
mifx_status launch_depth16_copy(hipStream_t s, Img in, Img out)
{
    const dim3 block(64, 4, 1);
    hipLaunchKernelGGL(depth16_copy_kernel, grid2d(out, block), block, 0, s, in, out);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}

// noise planes with a null `sobol`: C2 / C3 only
mifx_status launch_postfx_prep(hipStream_t s, Img depth, Img motion, Img reproj, Img closest, const CamK& cur, const CamK& prev, const uint8_t* sobol, const uint8_t* tile, Img noiseXY,
                               Img noiseZW, uint32_t frame, bool halfPrecisionDepth)
{
    dim3 block(64, 4, 1);
    dim3 grid = grid2d(depth, block);
    const unsigned noiseRow0 = grid.y;
    const NoiseK noise{sobol, tile, noiseXY, noiseZW, frame};
    if (sobol != nullptr) grid.y += (64u + grid.x - 1u) / grid.x;
    if (cur.reversedDepth) hipLaunchKernelGGL(postfx_prep_kernel<true>, grid, block, 0, s, depth, motion, reproj, closest, cur, prev, noise, noiseRow0, halfPrecisionDepth ? 1 : 0);
    else hipLaunchKernelGGL(postfx_prep_kernel<false>, grid, block, 0, s, depth, motion, reproj, closest, cur, prev, noise, noiseRow0, halfPrecisionDepth ? 1 : 0);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
