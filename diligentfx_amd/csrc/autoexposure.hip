// autoexposure.hip -- average scene luminance for ToneMap(), after the reference's light-scattering post-process:
//   low-resolution luminance  UnwarpEpipolarScattering.fx:283-307 (no in-scattering / extinction) + GetWeightedLogLum, AtmosphereShadersCommon.fxh:197-203
//   mip chain to 1x1          IDeviceContext::GenerateMips on g_tex2DLowResLuminance (EpipolarLightScattering.cpp:2503): 2x2 box per level
//   UpdateAverageLuminancePS  UpdateAverageLuminance.fx:12-29, alpha-blended into the 1x1 average (.cpp:1827, BS_AlphaBlend)
// One workgroup of 1024 threads does all of it: thread t owns the 2x2 block of low-resolution texels with Morton index t, so that the box
// levels are pairs of wave shuffles (lane ^ 1 = x neighbour, lane ^ 2 = y neighbour, then ^4 / ^8, ^16 / ^32); one value per wave goes
// through LDS to the first wave for the last two levels.
#include "mifx_host.h"

namespace mifx
{
constexpr int kLowRes = 64; // 1 << (LOW_RES_LUMINANCE_MIPS - 1), AtmosphereShadersCommon.fxh:54-56

MIFX_D v2 weighted_log_lum(v3 color, float minLuminance) // GetWeightedLogLum :197-203
{
    const float luminance = dot(color, v3{0.212671f, 0.715160f, 0.072169f}); // RGB_TO_LUMINANCE :84
    const float lumWeight = saturate(fdiv(luminance - minLuminance, minLuminance));
    const float logLum    = m_log(fmaxf(luminance, 1e-5f));
    return v2{logLum * lumWeight, lumWeight};
}
MIFX_D unsigned compact_bits(unsigned v) // even bits of v, packed
{
    v &= 0x55555555u;
    v = (v | (v >> 1)) & 0x33333333u;
    v = (v | (v >> 2)) & 0x0f0f0f0fu;
    v = (v | (v >> 4)) & 0x00ff00ffu;
    return v;
}
MIFX_D v2 box4(v2 a, v2 b, v2 c, v2 d) { return v2{((a.x + b.x) + (c.x + d.x)) * 0.25f, ((a.y + b.y) + (c.y + d.y)) * 0.25f}; }
// one 2x2 box level across the lanes {l, l^s, l^2s, l^3s}: every lane of the group ends up with the average
MIFX_D v2 box_level_shuffle(v2 v, int s)
{
    v = v2{v.x + __shfl_xor(v.x, s), v.y + __shfl_xor(v.y, s)};
    v = v2{v.x + __shfl_xor(v.x, 2 * s), v.y + __shfl_xor(v.y, 2 * s)};
    return v2{v.x * 0.25f, v.y * 0.25f};
}

// one texel of the low-resolution luminance: UnwarpEpipolarScattering.fx:283-307
MIFX_D v2 low_res_luminance(const Img& color, int x, int y, int packed)
{
    const float u = (float(x) + 0.5f) * (1.0f / float(kLowRes)), v = (float(y) + 0.5f) * (1.0f / float(kLowRes));
    const v4    c = sample_linear_clamp_hdr(color, u, v, packed); // g_tex2DColorBuffer.SampleLevel(linear clamp, f2UV, 0)
    return weighted_log_lum(xyz(c), 0.01f);              // MinLumn = 0.01 (UnwarpEpipolarScattering.fx:305)
}

// SAMPLE: the low-resolution texels are taken from the colour buffer (and stored); otherwise they are read back from lowRes -- row-band sharding, where every
// rank samples the rows whose footprint lies in its band (autoexposure_rows_kernel), the rows are exchanged and every rank reduces the same 64x64 values.
template <bool SAMPLE> __global__ __launch_bounds__(1024) void autoexposure_kernel(Img color, Img lowRes, float* average, float elapsedTime, int lightAdaptation, int packed)
{
    __shared__ v2 perWave[16];
    const unsigned t  = threadIdx.x;
    const int      bx = int(compact_bits(t)), by = int(compact_bits(t >> 1)); // Morton -> (x, y) of the 2x2 block, 0..31
    v2 q[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
    {
        const int x = 2 * bx + (i & 1), y = 2 * by + (i >> 1);
        if (SAMPLE)
        {
            q[i] = low_res_luminance(color, x, y, packed);
            st<v2>(lowRes, x, y, q[i]);
        }
        else
            q[i] = ld<v2>(lowRes, x, y);
    }
    v2 v = box4(q[0], q[1], q[2], q[3]); // 32x32
    v = box_level_shuffle(v, 1);         // 16x16
    v = box_level_shuffle(v, 4);         //  8x8
    v = box_level_shuffle(v, 16);        //  4x4: one value per wave
    if ((t & 63u) == 0u) perWave[t >> 6] = v;
    __syncthreads();
    if (t < 64u)
    {
        v = perWave[t & 15u];
        v = box_level_shuffle(v, 1); // 2x2
        v = box_level_shuffle(v, 4); // 1x1 = g_tex2DLowResLuminance.Load(int3(0, 0, LOW_RES_LUMINANCE_MIPS - 1))
        if (t == 0u)
        {
            // UpdateAverageLuminancePS :17-28
            float newWeight = lightAdaptation ? 1.0f - m_exp(-1.0f * elapsedTime) : 1.0f; // fAdaptationRate = 1
            const float logLum = fdiv(v.x, fmaxf(v.y, 1e-6f));
            newWeight *= saturate(fdiv(v.y, 1e-3f));
            const float lum = m_exp(logLum);
            *average = lum * newWeight + *average * (1.0f - newWeight); // BS_AlphaBlend: src * a + dst * (1 - a)
        }
    }
}

// rows [row0, row0 + gridDim.x) of the low-resolution luminance, one workgroup (= one wave) per row
__global__ __launch_bounds__(64) void autoexposure_rows_kernel(Img color, Img lowRes, int row0, int packed)
{
    const int x = int(threadIdx.x), y = row0 + int(blockIdx.x);
    st<v2>(lowRes, x, y, low_res_luminance(color, x, y, packed));
}

mifx_status launch_autoexposure(hipStream_t s, Img color, Img lowRes, float* average, float elapsedTime, int lightAdaptation, bool packedIn)
{
    hipLaunchKernelGGL(autoexposure_kernel<true>, dim3(1, 1, 1), dim3(1024, 1, 1), 0, s, color, lowRes, average, elapsedTime, lightAdaptation, packedIn ? 1 : 0);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_autoexposure_rows(hipStream_t s, Img color, Img lowRes, int rowBegin, int rowEnd, bool packedIn)
{
    if (rowEnd <= rowBegin) return MIFX_OK;
    hipLaunchKernelGGL(autoexposure_rows_kernel, dim3(unsigned(rowEnd - rowBegin), 1, 1), dim3(kLowRes, 1, 1), 0, s, color, lowRes, rowBegin, packedIn ? 1 : 0);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_autoexposure_reduce(hipStream_t s, Img lowRes, float* average, float elapsedTime, int lightAdaptation)
{
    hipLaunchKernelGGL(autoexposure_kernel<false>, dim3(1, 1, 1), dim3(1024, 1, 1), 0, s, lowRes, lowRes, average, elapsedTime, lightAdaptation, 0);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
