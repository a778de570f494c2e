// ssao_ao.hip -- ScreenSpaceAmbientOcclusion pass A3 (the AO estimator itself; the other passes are in ssao.hip) (XeGTAO-style GTAO / HBAO / visibility-bitmask AO with temporal
// accumulation, history-fix resampling and spatial denoise).  Math follows
// Shaders/PostProcess/ScreenSpaceAmbientOcclusion/private/SSAO_*.fx; the host sequence is in api_ssao.cpp.
//
// All planes are fp32 (AO, history length, depth).  Bandwidth accounting per pass: SURVEY.md Appendix C.
#include <cstdlib>
#include <cstring>

#include "mifx_host.h"
#include "mifx_effects.h"

namespace mifx
{

#define SSAO_SLICE_COUNT 3
#define SSAO_SAMPLES_PER_SLICE 3
#define SSAO_MAX_MIP 4
#define M_PI_F 3.14159265358979f
#define M_HALF_PI_F 1.57079632679490f

// SSAO_Common.fxh:25-28
MIFX_D float geometry_weight(v3 centerPos, v3 tapPos, v3 centerNormal, float planeDistNorm)
{
    return saturate(1.0f - fabsf(dot(tapPos - centerPos, centerNormal)) * planeDistNorm);
}

// ------------------------------------------------------------------------------------------------ A3: ambient occlusion (SSAO_ComputeAmbientOcclusion.fx:40-236)
MIFX_D float fast_acos(float v) // :47-53
{
    float a = fabsf(v);
    float r = -0.156583f * a + M_HALF_PI_F;
    r *= fsqrt(1.0f - a);
    return (v >= 0.0f) ? r : M_PI_F - r;
}
// g_TexturePrefilteredDepth.SampleLevel(Sam_PointClamp, uv, mip): nearest mip, nearest texel, clamp addressing.  The camera-z twin of the pyramid lives in one
// allocation (mifx_ssao::camz_slab) read through a buffer resource, and the block keeps one 32-byte record per level in LDS: a tap is
//     x = floor(med3(u * w, 0, w - 1)), y likewise   (the clamp in the float domain: floor and clamp commute, and w - 1 is exact)
//     offset = level offset + y * pitch + x * 4      (v_mad_u32_u24 + v_lshl_add_u32)
// i.e. 8 vector instructions and a buffer_load_dword where the generic Img path takes 14 with a 64-bit multiply-add.
struct CamzLevel
{
    unsigned off, pitch;
    float    w, h, wm1, hm1;
    unsigned pad[2];
};
struct CamzLds
{
    __amdgpu_buffer_rsrc_t rsrc;
    const CamzLevel*       lv;
};
MIFX_D float sample_prefiltered_depth(const CamzLds& cz, int byteOffset, float u, float v) // byteOffset = level * sizeof(CamzLevel)
{
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    const char*   p = reinterpret_cast<const char*>(cz.lv) + byteOffset;
    const u32x2   a = *reinterpret_cast<const u32x2*>(p);
    const mifx_f4 r = *reinterpret_cast<const mifx_f4*>(p + 8);
    const int x = floor_to_int(__builtin_amdgcn_fmed3f(u * r.x, 0.0f, r.z));
    const int y = floor_to_int(__builtin_amdgcn_fmed3f(v * r.y, 0.0f, r.w));
    unsigned row, off;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(y), "v"(a.y), "v"(a.x)); // rows and pitch < 2^24
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(x), "v"(row));
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(cz.rsrc, int(off), 0, 0));
}
// Level of a tap `lenSq` squared pixels away: the reference evaluates floor(clamp(log2(length(offset)) - DepthMIPSamplingOffset, 0, 4) + 0.5)
// (SSAO_ComputeAmbientOcclusion.fx, SampleDepth); the level only changes where len crosses 2^(k + 0.5 + offset), i.e. where len^2 crosses t0 * 4^k with
// t0 = 2^(1 + 2 offset) -- no sqrt + log2 (a fifth of the kernel's instructions in round 1).  Positive floats order like their bit patterns and a factor 4 is
// +2 in the exponent field, so the number of thresholds at or below len^2 is floor((bits(len^2) - bits(t0)) / 2^24) + 1, clamped to the levels that exist:
// a subtraction, an arithmetic shift and a clamp instead of four comparisons and four adds.  Either form can disagree with the reference's only for a tap
// within one rounding error of a threshold.  Returns the level times sizeof(CamzLevel), the offset of its record.
MIFX_D int tap_mip_offset(float lenSq, int t0BitsBiased /* bits(t0) - 2^24 */, int levels)
{
    const int l = (int(__float_as_uint(lenSq)) - t0BitsBiased) >> 24;
    return clampi(l, 0, levels - 1) * int(sizeof(CamzLevel));
}
MIFX_D unsigned occluded_sectors(float minH, float maxH, unsigned bits) // :77-98
{
    minH = saturate(minH);
    maxH = saturate(maxH);
    unsigned result = bits;
    if (maxH > minH)
    {
        const unsigned sectors = 32u;
        unsigned startI = min(unsigned(minH * float(sectors)), sectors - 1u);
        unsigned endI   = min(unsigned(ceilf(maxH * float(sectors))), sectors);
        if (endI > startI)
        {
            unsigned angle = endI - startI;
            unsigned field = angle >= 32u ? 0xFFFFFFFFu : ((1u << angle) - 1u);
            result |= field << startI;
        }
    }
    return result;
}

// --- smooth tail of the GTAO / HBAO estimator (view-space reconstruction of the fetched taps, horizon cosines, arc integration).
// These are smooth functions of the fetched depths whose output (a visibility in [0,1]) moves by ~1e-7 absolute when an intermediate moves
// by an ulp, so divisions / square roots use the 1-ulp hardware rcp / sqrt / rsq instead of the ~12-instruction IEEE sequences (35 % of
// the kernel's ALU work).  Trigonometry stays libm (v_sin / v_cos are only ~1e-5 accurate: measured 1e-2 output error), and everything that
// decides WHICH texel / mip is fetched (slice direction, sample offsets, log2 of the pixel distance) stays on the strict path.
// fused multiply-adds for the same smooth tail: one rounding instead of two, and a dot product in 3 instructions instead of 5
MIFX_D float dot_fma(v3 a, v3 b) { return __builtin_fmaf(a.x, b.x, __builtin_fmaf(a.y, b.y, a.z * b.z)); }
MIFX_D float lerp_fma(float a, float b, float t) { return __builtin_fmaf(t, b - a, a); }
MIFX_D float fast_acos_q(float v)
{
    float a = fabsf(v);
    float r = -0.156583f * a + M_HALF_PI_F;
    r *= q_sqrt(1.0f - a);
    return (v >= 0.0f) ? r : M_PI_F - r;
}
#ifndef MIFX_A3_WAVES
#define MIFX_A3_WAVES 7
#endif
#ifdef MIFX_A3_CAP // experiment knob: at most this many waves per SIMD
#define MIFX_A3_OCC __attribute__((amdgpu_waves_per_eu(MIFX_A3_CAP, MIFX_A3_CAP)))
#else
#define MIFX_A3_OCC MIFX_WAVES(MIFX_A3_WAVES)
#endif
// EXPERIMENT (round 6, MIFX_A3_COPY_ROLE=k; never the shipped launch): the verdict's question "does the dispatcher refuse to mix two grids on a CU, or is the mixed CU
// slower?" needs ONE grid whose workgroups alternate roles.  ROLES = true: of every k + 1 consecutive workgroups of a block row, k run A3 and one streams its share of
// `copy.bytes` from copy.src to copy.dst, 16 bytes per lane and access (the composite's traffic: tools/exp_a3_copy_role.py times the mixed grid against A3 and the copy
// alone).  The A3 role's arithmetic and tap order are those of the shipped kernel (same body).
struct CopyRole
{
    const mifx_f4* src;
    mifx_f4*       dst;
    unsigned     perBlock; // 16-byte elements each copy workgroup moves (a multiple of 256 x MIFX_A3_COPY_DEPTH)
    unsigned     k;        // A3 workgroups per copy workgroup
    unsigned     copiesPerRow;
};
template <int ALGO, bool ROLES = false> __global__ __launch_bounds__(256) MIFX_A3_OCC void ssao_compute_ao_kernel(Pyr depthPyr, HizSlab camzSlab, Img normal, Img noiseZW, Img out, CamK cam, SsaoK k, CopyRole copy)
{
    int bx = int(blockIdx.x);
    if (ROLES)
    {
        const unsigned q = blockIdx.x / (copy.k + 1u), r = blockIdx.x - q * (copy.k + 1u);
        if (r == copy.k)
        {
            const size_t base = (size_t(blockIdx.y) * copy.copiesPerRow + q) * copy.perBlock + threadIdx.x;
#ifndef MIFX_A3_COPY_DEPTH
#define MIFX_A3_COPY_DEPTH 8 // independent 16-byte loads in flight per lane: 8 KB per wave (4: the first version of the experiment)
#endif
            for (unsigned i = 0; i < copy.perBlock; i += 256u * MIFX_A3_COPY_DEPTH)
            {
                mifx_f4 v[MIFX_A3_COPY_DEPTH];
#pragma unroll
                for (int j = 0; j < MIFX_A3_COPY_DEPTH; ++j) v[j] = __builtin_nontemporal_load(copy.src + base + i + 256u * unsigned(j));
#pragma unroll
                for (int j = 0; j < MIFX_A3_COPY_DEPTH; ++j) __builtin_nontemporal_store(v[j], copy.dst + base + i + 256u * unsigned(j));
            }
            return;
        }
        bx = int(q * copy.k + r);
    }
    // taps read the camera-z pyramid (A2 writes depth_to_camera_z of every level beside the depth pyramid): one division less per tap
    __shared__ CamzLevel camzLv[8];
    if (threadIdx.x < 8u)
    {
        const unsigned m = threadIdx.x;
        camzLv[m] = CamzLevel{camzSlab.offset[m], camzSlab.pitch[m], float(camzSlab.w[m]), float(camzSlab.h[m]), float(camzSlab.w[m]) - 1.0f, float(camzSlab.h[m]) - 1.0f, {0u, 0u}};
    }
    __syncthreads();
    const CamzLds camz{__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(camzSlab.base), 0, int(camzSlab.bytes), 0x00020000), camzLv};
    const int levels = camzSlab.levels;
    const int t0BitsBiased = int(__float_as_uint(k.MipLenSq[0])) - (1 << 24);
    int x, y;
    bool inside = ROLES ? tiled_xy_at(out, bx, x, y) : tiled_xy_xcd(out, x, y);
    if (!ROLES && (MIFX_ROWS_UP & 1024)) // (experiment bit: the tile rows from the last to the first)
    {
        y      = int(gridDim.y - 1u - blockIdx.y) * 8 + (int(threadIdx.x & 63u) >> 3) + out.y0;
        inside = x < out.w && y < row_end(out);
    }
    if (!inside) return;

    const v2 position{float(x) + 0.5f, float(y) + 0.5f};
    const v2 uv{position.x * (k.UvScale * cam.ivw), position.y * (k.UvScale * cam.ivh)}; // Position * GetInvViewportSize()
    const v3 positionSS{uv.x, uv.y, sample_point_clamp_f(depthPyr.l[0], uv.x, uv.y)};
    if (is_background(positionSS.z, cam.reversedDepth != 0))
    {
        st<ao_t>(out, x, y, 1.0f); // the reference discards and keeps the cleared value 1.0 (ScreenSpaceAmbientOcclusion.cpp:982-985)
        return;
    }
    // LoadNormalWS: point-clamp sample at uv
    const int nx = clampi(int(floorf(uv.x * float(normal.w))), 0, normal.w - 1), ny = clampi(int(floorf(uv.y * float(normal.h))), 0, normal.h - 1);
    const v3  normalVS = mul_dir(xyz(ld<v4>(normal, nx, ny)), cam.view);
    v3        positionVS = screen_xy_camz_to_view_space(uv.x, uv.y, sample_prefiltered_depth(camz, 0, uv.x, uv.y), cam.proj);
    positionVS = positionVS + normalVS * k.SelfOcclusionOffset * positionVS.z; // fix self-occlusion
    const v3 viewVS = -normalize(positionVS);
    const v2 xi     = ld<v2>(noiseZW, x & 127, y & 127);

    const float effectRadius = k.EffectRadius * k.RadiusMultiplier;
    const float falloffRange = k.EffectFalloffRange * effectRadius;
    const float falloffFrom  = effectRadius - falloffRange;
    const float falloffMul   = fdiv(-1.0f, falloffRange);
    const float falloffAdd   = fdiv(falloffFrom, falloffRange) + 1.0f;
    float       sampleRadius = 0.5f * effectRadius * cam.proj.m[0];
    if (cam.proj.m[15] == 0.0f) sampleRadius = fdiv(sampleRadius, positionVS.z); // perspective

    constexpr bool QUICK = ALGO != MIFX_SSAO_ALGORITHM_VBAO; // the bitmask variant thresholds its angles into 32 sectors: keep it strict
#ifndef MIFX_A3_VGPR_SCALES
#define MIFX_A3_VGPR_SCALES 0
#endif
#if MIFX_A3_VGPR_SCALES
    // Round 6, MEASURED AND NOT TAKEN (profiles/r06_ab_a3_vgpr_scales.txt: 209.7 us with, 209.4 us without, same box, 60 frames).  A full-rate vector instruction with a
    // scalar-register source issues at half rate in the microbenchmark (profiles/r03_valu_issue_rate.txt), and a slice of this loop holds 33 of them: the two projection
    // scales of the taps' view-space reconstruction (the residual step of fdiv_finite, twelve per slice), the viewport size and the level threshold.  With all five values
    // in vector registers (72 registers, still seven waves per SIMD, same operations on the same values) the loop's static issue cost drops from 649 to 625 units -- and the
    // kernel does not move: what DESIGN.md section 7 listed as the last untried instruction-level item is worth nothing here.
    m44 projV = cam.proj;
    asm volatile("" : "+v"(projV.m[0]), "+v"(projV.m[5]));
    float vwV = cam.vw, vhV = cam.vh; // (the tap's pixel offset, twice per sample)
    int   t0V = t0BitsBiased;         // (the tap's level, once per sample)
    asm volatile("" : "+v"(vwV), "+v"(vhV), "+v"(t0V));
#else
    const m44& projV = cam.proj;
    const float vwV = cam.vw, vhV = cam.vh;
    const int   t0V = t0BitsBiased;
#endif

    float visibility = 0.0f;
#ifdef MIFX_A3_UNROLL
#pragma unroll
#endif
    for (int slice = 0; slice < SSAO_SLICE_COUNT; ++slice)
    {
        // (Measured in round 3 and not taken: the three directions from one sine / cosine pair and two rotations by 60 degrees -- a third of the slice set-up, every parity
        //  case unchanged to the digit -- and one reciprocal square root per tap in place of the square root + reciprocal.  In a 10-frame rocprofv3 run, where the
        //  clocks are still ramping, the first looks like -9 %; over 60 frames the kernel and the frame take what they took: 0.2126 vs 0.212 ms, 1.809 vs 1.806 ms.
        //  profiles/r03_ab_a3_rotate.txt.  A3 is not issue-bound at the steady-state clock.)
        // (Round 4, measured and not taken: the addresses of all 18 taps of the three slices computed first and their loads issued as one group, the slice set-up then
        //  running under them -- bit-identical, 95 registers = 5 waves per SIMD, three times the loads in flight per SIMD: 215.4 us against 211.0 us for this loop
        //  (4 waves: 223.9, 6 waves with a 20-byte spill: 226.0; at this kernel's 7-wave hint the allocator serialises the loads again and spills: 280 us and 3x the
        //  HBM traffic; profiles/r04_ab_a3_batch.txt).  A3 does not wait for its taps: the vector L1 is busy 76 % of its time with their tag look-ups
        //  (tools/microbench/tcp_gather_rate.hip, profiles/r04_tcp_gather_rate.txt).)
        // (Round 4, measured and not taken: the taps served from LDS -- 512-thread workgroups on 32 x 16 pixel tiles copy the texels within 14 texels of the tile from each of the
        //  five levels first (the reference's level selection keeps a tap 7 .. 13.9 texels of its own level from the pixel), per-tap fall-back to memory, bit-identical:
        //  269 us against 215 us.  The kernel is ALSO at ~85 % of its VALU issue capacity and the window costs +27 % instructions: profiles/r04_ab_a3_window.txt.)
        const float phi = (xi.x + fdiv(float(slice), 3.0f)) * M_PI_F; // ComputeSliceDirection :40-45
        v2 omega;
        m_sincos(phi, omega.y, omega.x); // phi in [0, 5/3 pi)
        const v3    sliceDir{omega.x, omega.y, 0.0f};
        const v3    orthoSliceDir = sliceDir - dot(sliceDir, viewVS) * viewVS;
        const v3    axisRaw       = cross(sliceDir, viewVS);
        const v3    axis          = QUICK ? axisRaw * q_rsqrt(dot(axisRaw, axisRaw)) : normalize(axisRaw);
        const v3    projNormal    = normalVS - axis * dot(normalVS, axis);
        const float projNormalLen = QUICK ? q_sqrt(dot(projNormal, projNormal)) : length(projNormal);
        const float cosNorm       = QUICK ? saturate(dot(projNormal, viewVS) * q_rcp(projNormalLen)) : saturate(dot(projNormal / projNormalLen, viewVS));
        const float n             = signf(dot(orthoSliceDir, projNormal)) * (QUICK ? fast_acos_q(cosNorm) : fast_acos(cosNorm));

        unsigned occluded = 0u;
        const float sinN = QUICK ? m_sin_bounded(n) : m_sin(n); // |n| <= pi/2
        v2 minCos = QUICK ? v2{-sinN, sinN} /* == cos(n + pi/2), cos(n - pi/2) */ : v2{m_cos(n + M_HALF_PI_F), m_cos(n - M_HALF_PI_F)};
        v2 maxCos = minCos;

        v2 sampleDir{omega.x * 0.5f * sampleRadius, omega.y * -0.5f * sampleRadius}; // Omega * F3NDC_XYZ_TO_UVD_SCALE.xy * SampleRadius
        sampleDir.x *= cam.vh * cam.ivw;                                             // aspect-ratio correction

        for (int si = 0; si < SSAO_SAMPLES_PER_SLICE; ++si)
        {
            const float noise  = fracf(xi.y + float(slice + si * SSAO_SAMPLES_PER_SLICE) * 0.6180339887498948482f);
            const float sample = fdiv(float(si) + noise, float(SSAO_SAMPLES_PER_SLICE));
            const v2    offset = sample * sample * sampleDir;
            const v2    p0{positionSS.x + offset.x, positionSS.y + offset.y};
            const v2    p1{positionSS.x - offset.x, positionSS.y - offset.y};
            const v2    offPx{offset.x * vwV, offset.y * vhV};
            const int   mip = tap_mip_offset(dot(offPx, offPx), t0V, levels);
            const float z0 = sample_prefiltered_depth(camz, mip, p0.x, p0.y), z1 = sample_prefiltered_depth(camz, mip, p1.x, p1.y);
            // (the reconstruction itself stays bit-exact: d = s - positionVS is a cancelling difference for nearby taps.  Measured in round 2: multiplying by the
            //  reciprocals of the two projection scales instead of dividing -- an equally accurate rounding -- moved 0.2-0.7 % of the AO texels by up to 1e-2: the
            //  horizon angle is acos of a cosine that approaches 1 for taps beside the centre, where one ulp of the position is amplified without bound.)
            const v3 s0 = screen_xy_camz_to_view_space(p0.x, p0.y, z0, projV);
            const v3 s1 = screen_xy_camz_to_view_space(p1.x, p1.y, z1, projV);

            if (ALGO == MIFX_SSAO_ALGORITHM_VBAO)
            {
                // ComputeSampleOcclusion :100-119
                const v3 d0 = s0 - positionVS, d1 = s1 - positionVS;
                const v3 thick = viewVS * k.BitmaskThickness;
                const v2 w{saturate(length(d0) * falloffMul + falloffAdd), saturate(length(d1) * falloffMul + falloffAdd)};
                v4 fb{fast_acos(dot(normalize(d0), viewVS)), fast_acos(dot(normalize(d0 - thick), viewVS)), fast_acos(dot(normalize(d1), viewVS)),
                      fast_acos(dot(normalize(d1 - thick), viewVS))};
                const float nb = -n;
                fb = v4{saturate(fdiv(-fb.x - nb + M_HALF_PI_F, M_PI_F)), saturate(fdiv(-fb.y - nb + M_HALF_PI_F, M_PI_F)), saturate(fdiv(fb.z - nb + M_HALF_PI_F, M_PI_F)),
                        saturate(fdiv(fb.w - nb + M_HALF_PI_F, M_PI_F))};
                if (w.x > 0.0f) occluded = occluded_sectors(fb.y, fb.x, occluded);
                if (w.y > 0.0f) occluded = occluded_sectors(fb.z, fb.w, occluded);
            }
            else
            {
                // ComputeSampleHorizons :121-130
                const v3 d0 = s0 - positionVS, d1 = s1 - positionVS;
                const v2 dist{q_sqrt(dot_fma(d0, d0)), q_sqrt(dot_fma(d1, d1))};
                const v2 cosH{dot_fma(d0, viewVS) * q_rcp(dist.x), dot_fma(d1, viewVS) * q_rcp(dist.y)};
                const v2 w{saturate(__builtin_fmaf(dist.x, falloffMul, falloffAdd)), saturate(__builtin_fmaf(dist.y, falloffMul, falloffAdd))};
                maxCos = v2{fmaxf(maxCos.x, lerp_fma(minCos.x, cosH.x, w.x)), fmaxf(maxCos.y, lerp_fma(minCos.y, cosH.y, w.y))};
            }
        }

        if (ALGO == MIFX_SSAO_ALGORITHM_VBAO)
        {
            visibility += 1.0f - float(__popc(occluded)) / 32.0f;
        }
        else if (ALGO == MIFX_SSAO_ALGORITHM_HBAO)
        {
            const float hx = +fast_acos_q(maxCos.x), hy = -fast_acos_q(maxCos.y);
            visibility += 0.5f * (1.0f - m_cos_bounded(hx) + (1.0f - m_cos_bounded(hy))); // IntegrateArcUniform :55-58
        }
        else
        {
            const float hx = +fast_acos_q(maxCos.x), hy = -fast_acos_q(maxCos.y);
            // IntegrateArcCosWeighted :60-66
            const float h1 = hx * 2.0f, h2 = hy * 2.0f;
            visibility += projNormalLen * (0.25f * ((-m_cos_bounded(h1 - n) + cosNorm + h1 * sinN) + (-m_cos_bounded(h2 - n) + cosNorm + h2 * sinN)));
        }
    }
    st<ao_t>(out, x, y, fdiv(visibility, float(SSAO_SLICE_COUNT)));
}

static const dim3 kBlock(64, 4, 1);

mifx_status launch_ssao_compute_ao(hipStream_t s, const Pyr& depthPyr, const Pyr& camzPyr, Img normal, Img noiseZW, Img out, const CamK& cam, const mifx_ssao_attribs& a,
                                   bool halfResolution, bool halfPrecisionDepth)
{
#ifndef MIFX_A3_BLOCK
#define MIFX_A3_BLOCK 256 // (measured: 64-thread workgroups, one 8x8 tile each, are 15 % slower)
#endif
    const dim3 grid(xcd_grid_x((out.w + MIFX_A3_BLOCK / 8 - 1) / (MIFX_A3_BLOCK / 8)), (window_rows(out) + 7) / 8, 1), kTiled(MIFX_A3_BLOCK, 1, 1);
    const SsaoK k = make_k(a, halfResolution, halfPrecisionDepth);
    // the levels of the camera-z pyramid as offsets from their lowest address (mifx_ssao allocates them as one slab)
    HizSlab camzSlab{};
    MIFX_REQUIRE(camzPyr.levels >= 1 && camzPyr.levels <= 8, "camera-z pyramid: %d levels", camzPyr.levels);
    const unsigned char* lo = camzPyr.l[0].p;
    const unsigned char* hi = lo;
    for (int i = 0; i < camzPyr.levels; ++i)
    {
        const unsigned char* b = camzPyr.l[i].p;
        const unsigned char* e = b + size_t(camzPyr.l[i].pitch) * size_t(camzPyr.l[i].h);
        lo = b < lo ? b : lo;
        hi = e > hi ? e : hi;
    }
    MIFX_REQUIRE(size_t(hi - lo) < (size_t(1) << 31), "camera-z pyramid: the levels span %zu bytes; they must share one allocation below 2 GiB", size_t(hi - lo));
    camzSlab.base   = lo;
    camzSlab.bytes  = uint32_t(hi - lo);
    camzSlab.levels = camzPyr.levels;
    for (int i = 0; i < 8; ++i)
    {
        const Img& l = camzPyr.l[i < camzPyr.levels ? i : camzPyr.levels - 1];
        MIFX_REQUIRE(l.w < (1 << 24) && l.h < (1 << 24) && l.pitch < (1 << 24), "camera-z pyramid: level %d too large", i);
        camzSlab.offset[i] = uint32_t(l.p - lo); camzSlab.pitch[i] = uint32_t(l.pitch); camzSlab.w[i] = uint32_t(l.w); camzSlab.h[i] = uint32_t(l.h);
    }
    static const unsigned ldsPad = occupancy_pad_from_env("MIFX_A3_LDS_PAD"); // experiment knob, see launch_ssr_intersection
    const CopyRole none{};
    if (const char* e = std::getenv("MIFX_A3_COPY_ROLE")) // EXPERIMENT (see CopyRole): "k" or "k:megabytes" -- one copy workgroup per k A3 workgroups moves `megabytes` (default 847 = the composite's traffic, read + write)
    {
        const unsigned kk = unsigned(std::atoi(e));
        if (kk >= 1u && a.Algorithm == MIFX_SSAO_ALGORITHM_GTAO && MIFX_A3_BLOCK == 256)
        {
            const char*  colon = std::strchr(e, ':');
            const double mb    = colon ? std::atof(colon + 1) : 847.0;
            CopyRole c{};
            c.k            = kk;
            c.copiesPerRow = (grid.x + kk - 1u) / kk; // (a trailing partial group gets a copy workgroup as well; its missing A3 workgroups fall outside the image)
            const size_t nCopy = size_t(c.copiesPerRow) * grid.y;
            size_t per = size_t(mb * 0.5e6 / 16.0 / double(nCopy)); // 16-byte elements per copy workgroup (half of the traffic is read, half written)
            per        = (per + 256u * MIFX_A3_COPY_DEPTH - 1u) / (256u * MIFX_A3_COPY_DEPTH) * (256u * MIFX_A3_COPY_DEPTH);
            c.perBlock = unsigned(per);
            static void*  scratch = nullptr;
            static size_t scratchBytes = 0;
            const size_t need = per * nCopy * 16u;
            if (scratchBytes < 2u * need)
            {
                if (scratch) (void)hipFree(scratch);
                MIFX_HIP_CHECK(hipMalloc(&scratch, 2u * need));
                MIFX_HIP_CHECK(hipMemset(scratch, 0, 2u * need));
                scratchBytes = 2u * need;
            }
            c.src = static_cast<const mifx_f4*>(scratch);
            c.dst = reinterpret_cast<mifx_f4*>(static_cast<unsigned char*>(scratch) + need);
            const dim3 g2(c.copiesPerRow * (kk + 1u), grid.y, 1);
            hipLaunchKernelGGL((ssao_compute_ao_kernel<MIFX_SSAO_ALGORITHM_GTAO, true>), g2, kTiled, ldsPad, s, depthPyr, camzSlab, normal, noiseZW, out, cam, k, c);
            MIFX_HIP_CHECK(hipGetLastError());
            return MIFX_OK;
        }
    }
    switch (a.Algorithm)
    {
        case MIFX_SSAO_ALGORITHM_GTAO: hipLaunchKernelGGL((ssao_compute_ao_kernel<MIFX_SSAO_ALGORITHM_GTAO>), grid, kTiled, ldsPad, s, depthPyr, camzSlab, normal, noiseZW, out, cam, k, none); break;
        case MIFX_SSAO_ALGORITHM_HBAO: hipLaunchKernelGGL((ssao_compute_ao_kernel<MIFX_SSAO_ALGORITHM_HBAO>), grid, kTiled, ldsPad, s, depthPyr, camzSlab, normal, noiseZW, out, cam, k, none); break;
        case MIFX_SSAO_ALGORITHM_VBAO: hipLaunchKernelGGL((ssao_compute_ao_kernel<MIFX_SSAO_ALGORITHM_VBAO>), grid, kTiled, ldsPad, s, depthPyr, camzSlab, normal, noiseZW, out, cam, k, none); break;
        default: set_error("unknown SSAO algorithm %u", a.Algorithm); return MIFX_ERR_INVALID_ARG;
    }
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
