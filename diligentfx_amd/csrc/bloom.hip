// bloom.hip -- Bloom (B1-B3); TemporalAntiAliasing (T1) is in taa.hip (its own file so that the two can have different contraction policies, build.py FMA_SOURCES).
//   B1 Shaders/PostProcess/Bloom/private/Bloom_ComputePrefilteredTexture.fx:19-85   (Karis-weighted 13-tap downsample + soft-knee threshold)
//   B2 .../Bloom_ComputeDownsampledTexture.fx:11-44                                 (13-tap downsample)
//   B3 .../Bloom_ComputeUpsampledTexture.fx:20-55                                   (3x3 tent upsample + add; final composite when uInstID != 0)
//   B3 final + M2: in the chain the copy-frame pass ToneMap() (tonemap.hip, ToneMapping.fxh:87-226) is the tail of the final up-sample (bloom_final_tonemap_kernel)
// Texture-unit behaviour is reproduced in software with exact fp32 weights: B1/B2 sample with a linear BORDER sampler
// (Bloom.cpp:52-59,185,219; border colour 0), B3 and the TAA history with linear CLAMP (Bloom.cpp:253-254, TemporalAntiAliasing.cpp:234).
#include "mifx_host.h"
#include "mifx_tonemap.h"

namespace mifx
{
// ------------------------------------------------------------------------------------------------ LDS-staged texel source
// Bloom's spatial filters re-read every source texel many times (13 bilinear taps = 52 fetches per output texel in B1/B2, 36 in B3).
// A workgroup therefore stages the footprint of its 32x8 output block in LDS once (coalesced float4 loads, the addressing mode applied at
// fill time) and the taps read the tile.  The arithmetic -- weights, accumulation order -- is exactly that of the global-memory path, so
// results are bit-identical.  The launcher only selects the staged kernel when tile_fits() proves that every tap of the block lands in
// the tile (one texel of slack per side for the fp32 rounding of the tap position), so fetch() is a plain LDS read: no bounds test, no
// fallback branch (an earlier per-fetch fallback cost 6 scalar instructions x 36-52 fetches per wave and made B3 issue-bound).
// TAG: the storage type of the source plane (v4: a colour plane; bloom_t: a level of the Bloom pyramid); the tile itself always holds fp32 texels
// LT: the texel kept in LDS -- v4, or v3 for the filters that read rgb only (B1 / B2: a quarter less LDS per workgroup and per fetch)
template <int TW, int TH, bool BORDER, class TAG = v4, class LT = v4> struct Tile
{
    static constexpr bool kZeroOutside = BORDER; // BORDER tiles hold 0 for out-of-image texels: samplers need no in-image test
    LT* lds;       // TW * TH texels
    Img im;
    int x0, y0;    // image coordinates of tile texel (0, 0)
    static constexpr bool border = BORDER; // true: out-of-image texels are 0 (BORDER addressing); false: coordinates are clamped (CLAMP addressing)

    // Round 5: all of a thread's texels are requested before the first is written.  The fill used to be a loop of  load - wait - write  per texel (one load in flight:
    // tools/isa_roundtrips.py), i.e. seven dependent round trips for the 72 x 24 tile of B1 / B2 in front of the barrier -- on kernels whose whole life is ~ten.
    // NT = the block's thread count (32 x 8); a thread's surplus slots repeat its first texel (no branch between the loads) and are not written.
#ifndef MIFX_BLOOM_FILL_ROLLED
    template <int NT = 256> MIFX_D void fill() const
    {
        constexpr int N = (TW * TH + NT - 1) / NT;
        const int tid = threadIdx.y * blockDim.x + threadIdx.x;
        v4 v[N];
#pragma unroll
        for (int k = 0; k < N; ++k)
        {
            const int i  = tid + k * NT, ii = i < TW * TH ? i : tid % (TW * TH);
            const int gx = x0 + ii % TW, gy = y0 + ii / TW;
            const v4  t  = ld<TAG>(im, clampi(gx, 0, im.w - 1), clampi(gy, 0, im.h - 1));
            v[k] = t;
        }
#pragma unroll
        for (int k = 0; k < N; ++k) keep_here(v[k]); // (every request is out before the first value is touched)
#pragma unroll
        for (int k = 0; k < N; ++k)
        {
            const int i  = tid + k * NT;
            const int gx = x0 + i % TW, gy = y0 + i / TW;
            if (i < TW * TH) lds[i] = to_lds((!border || (gx >= 0 && gy >= 0 && gx < im.w && gy < im.h)) ? v[k] : mk4(0.0f), static_cast<const LT*>(nullptr));
        }
    }
#else
    MIFX_D void fill() const // (the form of rounds 1-4, for A/B builds)
    {
        const int tid = threadIdx.y * blockDim.x + threadIdx.x, nthreads = blockDim.x * blockDim.y;
        for (int i = tid; i < TW * TH; i += nthreads)
        {
            const int tx = i % TW, ty = i / TW, gx = x0 + tx, gy = y0 + ty;
            v4 v = mk4(0.0f);
            if (border) { if (gx >= 0 && gy >= 0 && gx < im.w && gy < im.h) v = ld<TAG>(im, gx, gy); }
            else v = ld<TAG>(im, clampi(gx, 0, im.w - 1), clampi(gy, 0, im.h - 1));
            lds[i] = to_lds(v, static_cast<const LT*>(nullptr));
        }
    }
#endif
    static MIFX_D v4 to_lds(v4 v, const v4*) { return v; }
    static MIFX_D v3 to_lds(v4 v, const v3*) { return xyz(v); }
    static MIFX_D v4 from_lds(v4 v) { return v; }
    static MIFX_D v4 from_lds(v3 v) { return mk4(v, 0.0f); }
    // texel at image coordinates (x, y); for CLAMP tiles (x, y) is already clamped by the caller, for BORDER tiles it may lie outside the image
    MIFX_D v4 fetch(int x, int y) const
    {
        return from_lds(lds[(y - y0) * TW + (x - x0)]);
    }
};
template <class TAG = v4> struct Direct // un-staged source with the same interface (fallback for shapes whose footprint does not fit the tile)
{
    static constexpr bool kZeroOutside = false;
    Img im;
    MIFX_D v4 fetch(int x, int y) const { return ld<TAG>(im, x, y); }
};
// the value a consumer of the Bloom output reads: an R11G11B10_FLOAT target in the reference (Bloom.cpp:137) -- three unsigned small floats, alpha reads as 1.  The
// native-storage build stores the output plane in that format (since round 4; the tone map and the auto exposure take it: mifx_device.h ld_hdr); the fused tone map
// below works on exactly these values.
#ifdef MIFX_STORAGE_H4
MIFX_D v4 bloom_output_value(v4 v) { return quantize_bloom(v); }
#else
MIFX_D v4 bloom_output_value(v4 v) { return v; }
#endif

// ------------------------------------------------------------------------------------------------ samplers (exact fp32 bilinear weights)
template <class SRC> MIFX_D v3 sample_linear_border_rgb(const SRC& src, int w, int h, float u, float v)
{
    const float fx = u * float(w) - 0.5f, fy = v * float(h) - 0.5f;
    const float x0f = floorf(fx), y0f = floorf(fy);
    const float wx = fx - x0f, wy = fy - y0f;
    const int   x0 = int(x0f), y0 = int(y0f);
    const float wgt[4] = {(1.0f - wx) * (1.0f - wy), wx * (1.0f - wy), (1.0f - wx) * wy, wx * wy};
    v3 acc = mk3(0.0f);
#pragma unroll
    for (int t = 0; t < 4; ++t)
    {
        const int x = x0 + (t & 1), y = y0 + (t >> 1);
        if (SRC::kZeroOutside || (x >= 0 && y >= 0 && x < w && y < h))
        {
            MIFX_FMA_BLOCK
            const v4 c = src.fetch(x, y);
            acc = v3{acc.x + c.x * wgt[t], acc.y + c.y * wgt[t], acc.z + c.z * wgt[t]};
        }
    }
    return acc;
}
template <class SRC> MIFX_D v3 sample_linear_clamp_rgb(const SRC& src, int w, int h, float u, float v)
{
    const Bilinear b = bilinear_uc(u * float(w), v * float(h), w, h);
    const v4 t00 = src.fetch(b.x0, b.y0), t10 = src.fetch(b.x1, b.y0), t01 = src.fetch(b.x0, b.y1), t11 = src.fetch(b.x1, b.y1);
    {
        MIFX_FMA_BLOCK
        return v3{t00.x * b.w00 + t10.x * b.w10 + t01.x * b.w01 + t11.x * b.w11, t00.y * b.w00 + t10.y * b.w10 + t01.y * b.w01 + t11.y * b.w11,
                  t00.z * b.w00 + t10.z * b.w10 + t01.z * b.w01 + t11.z * b.w11};
    }
}

// 13-tap pattern shared by B1 and B2
struct Taps13 { v3 A, B, C, D, E, F, G, H, I, J, K, L, M; };
template <class SRC> MIFX_D Taps13 fetch13(const SRC& src, int w, int h, v2 uv)
{
    const v2 ts{fdiv(1.0f, float(w)), fdiv(1.0f, float(h))};
    auto S = [&](float ox, float oy) { v3 r = sample_linear_border_rgb(src, w, h, uv.x + ts.x * ox, uv.y + ts.y * oy); MIFX_TAP_FENCE(r); return r; };
    Taps13 t;
    t.A = S(-2.0f, +2.0f); t.B = S(+0.0f, +2.0f); t.C = S(+2.0f, +2.0f);
    t.D = S(-2.0f, +0.0f); t.E = S(+0.0f, +0.0f); t.F = S(+2.0f, +0.0f);
    t.G = S(-2.0f, -2.0f); t.H = S(+0.0f, -2.0f); t.I = S(+2.0f, -2.0f);
    t.J = S(-1.0f, +1.0f); t.K = S(+1.0f, +1.0f); t.L = S(-1.0f, -1.0f); t.M = S(+1.0f, -1.0f);
    return t;
}
MIFX_D v2 pixel_uv(int x, int y, int w, int h) // NormalizedDeviceXYToTexUV(f2NormalizedXY) of the texel centre
{
    const v2 ndc{2.0f * fdiv(float(x) + 0.5f, float(w)) - 1.0f, 1.0f - 2.0f * fdiv(float(y) + 0.5f, float(h))};
    return ndc_to_uv(ndc);
}

// Workgroup = 32x8 output texels.  Footprint of the block in source texels for taps at uv +- `reach` source texels (bilinear => +1):
//   [floor((bx0 + 0.5) * sw / ow - 0.5 - reach) - 1,  floor((bx0 + 31.5) * sw / ow - 0.5 + reach) + 2]   (one texel of slack each side)
constexpr int kBX = 32, kBY = 8;
constexpr int kDownTW = 72, kDownTH = 24; // 2:1 reduction, reach 2: 31 * 2 + 8 = 70 (+ slack for odd source sizes)
constexpr int kUpTW = 24, kUpTH = 12;     // 1:2 magnification, reach 1: 31 / 2 + 6 = 22
MIFX_HD int tile_origin(int b0, int srcN, int outN, float reach) { return int(floorf((float(b0) + 0.5f) * float(srcN) / float(outN) - 0.5f - reach)) - 1; }
inline bool tile_fits(int srcN, int outN, int blockN, float reach, int tileN)
{
    // the widest footprint over all block positions is bounded by blockN * ratio + 2 * reach + 5
    return float(blockN - 1) * float(srcN) / float(outN) + 2.0f * reach + 5.0f <= float(tileN);
}

typedef v3 UpLds;   // B3 likewise (final pass 96.8 -> 93.1 us)
typedef v3 DownLds; // B1 / B2 read rgb only: 12-byte tile texels (20.7 instead of 27.6 KB per workgroup, ds_read_b96 fetches): B1 65.6 -> 55.4 us at 4K, same values
// ------------------------------------------------------------------------------------------------ B1
// SRC: the texel type of the source -- v4 (the TAA output), or bloom_t when depth of field precedes Bloom: its output is an R11G11B10_FLOAT target like Bloom's own
// (DepthOfField.cpp:281-289), a 4-byte plane in the native-storage build (the same type as v4 in the fp32 build)
template <bool STAGED, class SRC> __global__ __launch_bounds__(256) void bloom_prefilter_kernel(Img in, Img out, float threshold, float softThreshold)
{
    __shared__ DownLds lds[STAGED ? kDownTW * kDownTH : 1];
    const int by0 = int(block_row<4>()) * kBY + out.y0; // first row of this block (row window of `out`)
    const int x = blockIdx.x * kBX + threadIdx.x, y = by0 + int(threadIdx.y);
    Taps13 t;
    if (STAGED)
    {
        const Tile<kDownTW, kDownTH, true, SRC, DownLds> tile{lds, in, tile_origin(blockIdx.x * kBX, in.w, out.w, 2.0f), tile_origin(by0, in.h, out.h, 2.0f)};
        tile.fill();
        __syncthreads();
        if (x >= out.w || y >= row_end(out)) return;
        t = fetch13(tile, in.w, in.h, pixel_uv(x, y, out.w, out.h));
    }
    else
    {
        if (x >= out.w || y >= row_end(out)) return;
        t = fetch13(Direct<SRC>{in}, in.w, in.h, pixel_uv(x, y, out.w, out.h));
    }
    const float weights[5] = {0.125f, 0.125f, 0.125f, 0.125f, 0.5f};
    const v3 groups[5] = {(t.A + t.B + t.D + t.E) / 4.0f, (t.B + t.C + t.E + t.F) / 4.0f, (t.D + t.E + t.G + t.H) / 4.0f, (t.E + t.F + t.H + t.I) / 4.0f,
                          (t.J + t.K + t.L + t.M) / 4.0f};
    v4 sum = mk4(0.0f);
#pragma unroll
    for (int g = 0; g < 5; ++g)
    {
        const float w = weights[g] * fdiv(1.0f, 1.0f + luminance601(groups[g])); // KarisAverage :19-22
        sum += mk4(groups[g], 1.0f) * w;
    }
    const v3 color = xyz(sum) / (sum.w + 1.0e-5f);
    // Prefilter :24-35
    const float brightness = max_comp(color);
    const float knee = threshold * softThreshold;
    float soft = brightness - threshold + knee;
    soft = clampf(soft, 0.0f, 2.0f * knee);
    soft = fdiv(soft * soft * 0.25f, knee + 1.0e-5f);
    float contribution = fmaxf(soft, brightness - threshold);
    contribution = fdiv(contribution, fmaxf(brightness, 1.0e-5f));
    st<bloom_t>(out, x, y, mk4(color * contribution, 0.0f));
}

// ------------------------------------------------------------------------------------------------ B2
template <bool STAGED> __global__ __launch_bounds__(256) void bloom_downsample_kernel(Img in, Img out)
{
    __shared__ DownLds lds[STAGED ? kDownTW * kDownTH : 1];
    const int by0 = int(blockIdx.y) * kBY + out.y0; // first row of this block (row window of `out`)
    const int x = blockIdx.x * kBX + threadIdx.x, y = by0 + int(threadIdx.y);
    Taps13 t;
    if (STAGED)
    {
        const Tile<kDownTW, kDownTH, true, bloom_t, DownLds> tile{lds, in, tile_origin(blockIdx.x * kBX, in.w, out.w, 2.0f), tile_origin(by0, in.h, out.h, 2.0f)};
        tile.fill();
        __syncthreads();
        if (x >= out.w || y >= row_end(out)) return;
        t = fetch13(tile, in.w, in.h, pixel_uv(x, y, out.w, out.h));
    }
    else
    {
        if (x >= out.w || y >= row_end(out)) return;
        t = fetch13(Direct<bloom_t>{in}, in.w, in.h, pixel_uv(x, y, out.w, out.h));
    }
    v3 c = mk3(0.0f);
    c += (t.A + t.C + t.G + t.I) * 0.03125f;
    c += (t.B + t.D + t.F + t.H) * 0.0625f;
    c += (t.E + t.J + t.K + t.L + t.M) * 0.125f;
    st<bloom_t>(out, x, y, mk4(c, 0.0f));
}

// ------------------------------------------------------------------------------------------------ B3
// The 3x3 tent of bilinear samples (weights 1/16 2/16 1/16 / 2/16 4/16 2/16 / 1/16 2/16 1/16 = [1/4 1/2 1/4] x [1/4 1/2 1/4]) touches a 4x4
// texel block whenever the three tap positions of an axis fall into consecutive texel intervals (always for the 2:1 pyramid; checked per
// pixel).  Folding the tent into the bilinear weights per axis -- with the fractions f_k evaluated exactly as the reference does, fp32
// noise included -- gives 4 + 4 axis weights, and the result is the weighted sum of 16 texels instead of 9 x 4: the same products
// t_kx t_ky w_x w_y T summed in a different order (all terms non-negative: relative difference ~1e-7), 48 FMAs instead of 9 x (4 + 12).
struct TentAxis
{
    int   i[4];
    float w[4];
    bool  regular;
};
MIFX_D TentAxis tent_axis(float u, float ts, int n)
{
    const float l0 = (u + ts * -1.0f) * float(n) - 0.5f, l1 = (u + ts * 0.0f) * float(n) - 0.5f, l2 = (u + ts * 1.0f) * float(n) - 0.5f;
    const float f0 = floorf(l0), f1 = floorf(l1), f2 = floorf(l2);
    const float x0 = l0 - f0, x1 = l1 - f1, x2 = l2 - f2;
    TentAxis a;
    a.regular = (f1 == f0 + 1.0f) && (f2 == f0 + 2.0f);
    const int b = int(f0);
#pragma unroll
    for (int s = 0; s < 4; ++s) a.i[s] = clampi(b + s, 0, n - 1);
    a.w[0] = 0.25f * (1.0f - x0);
    a.w[1] = 0.25f * x0 + 0.5f * (1.0f - x1);
    a.w[2] = 0.5f * x1 + 0.25f * (1.0f - x2);
    a.w[3] = 0.25f * x2;
    return a;
}
// the up-sample of one output texel (x, y); false: the texel lies outside the image / the row window (after the block's LDS fill)
template <bool FINAL, bool STAGED> MIFX_D bool bloom_upsample_texel(UpLds* lds, Img input, Img down, Img out, float intensity, float alphaInterp, int srcPacked, int& x, int& y, v4& result)
{
    const int by0 = int(blockIdx.y) * kBY + out.y0; // first row of this block (row window of `out`)
    x = blockIdx.x * kBX + threadIdx.x;
    y = by0 + int(threadIdx.y);
    const Tile<kUpTW, kUpTH, false, bloom_t, UpLds> tile{lds, down, tile_origin(blockIdx.x * kBX, down.w, out.w, 1.0f), tile_origin(by0, down.h, out.h, 1.0f)};
    if (STAGED)
    {
        tile.fill();
        __syncthreads();
    }
    if (x >= out.w || y >= row_end(out)) return false;
    const v2 uv = pixel_uv(x, y, out.w, out.h);
    const v2 ts{fdiv(1.0f, float(down.w)), fdiv(1.0f, float(down.h))};
    auto S = [&](float ox, float oy) {
        v3 r = STAGED ? sample_linear_clamp_rgb(tile, down.w, down.h, uv.x + ts.x * ox, uv.y + ts.y * oy)
                            : sample_linear_clamp_rgb(Direct<bloom_t>{down}, down.w, down.h, uv.x + ts.x * ox, uv.y + ts.y * oy);
        MIFX_TAP_FENCE(r);
        return r;
    };
    v3 sum;
    const TentAxis ax = tent_axis(uv.x, ts.x, down.w), ay = tent_axis(uv.y, ts.y, down.h);
    if (STAGED && ax.regular && ay.regular)
    {
        sum = mk3(0.0f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            v3 row = mk3(0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                MIFX_FMA_BLOCK
                const v4 t = tile.fetch(ax.i[i], ay.i[j]);
                row = v3{row.x + t.x * ax.w[i], row.y + t.y * ax.w[i], row.z + t.z * ax.w[i]};
            }
            {
                MIFX_FMA_BLOCK
                sum = v3{sum.x + row.x * ay.w[j], sum.y + row.y * ay.w[j], sum.z + row.z * ay.w[j]};
            }
            MIFX_TAP_FENCE(sum);
        }
    }
    else
    {
        const v3 A = S(-1.0f, +1.0f), B = S(+0.0f, +1.0f), C = S(+1.0f, +1.0f);
        const v3 D = S(-1.0f, +0.0f), E = S(+0.0f, +0.0f), F = S(+1.0f, +0.0f);
        const v3 G = S(-1.0f, -1.0f), H = S(+0.0f, -1.0f), I = S(+1.0f, -1.0f);
        sum = E * 0.25f;
        sum += (B + D + F + H) * 0.125f;
        sum += (A + C + G + I) * 0.0625f;
    }
    // g_TextureInput has the resolution of the render target, so the linear-clamp sample at the texel centre IS the texel (the reference's
    // fp32 weights are 1 - O(1e-5); a direct load is the exact value)
    // (round 5, measured and not taken: this load requested first, beside the tile's texels, instead of here behind the filter -- one dependent round trip less on
    //  paper, 79.9 -> 84.7 us for the final pass: four more registers held across the filter)
    const v4 src4 = FINAL ? ld_hdr_once(input, x, y, srcPacked) : ld<bloom_t>(input, x, y); // final pass: the frame (srcPacked: depth of field's 4-byte output); otherwise the down-sampled level of this size
    const v3 src  = xyz(src4);
    result = FINAL ? bloom_output_value(mk4(lerp3(src, src + intensity * sum, alphaInterp), src4.w)) // alpha: pass-through of the input texel (fp32 build)
                   : mk4(src + sum, 0.0f);
    return true;
}
template <bool FINAL, bool STAGED> __global__ __launch_bounds__(256) void bloom_upsample_kernel(Img input, Img down, Img out, float intensity, float alphaInterp, int srcPacked)
{
    __shared__ UpLds lds[STAGED ? kUpTW * kUpTH : 1];
    int x, y;
    v4  r;
    if (bloom_upsample_texel<FINAL, STAGED>(lds, input, down, out, intensity, alphaInterp, srcPacked, x, y, r))
    {
        st<bloom_t>(out, x, y, r); // (FINAL: the output target, R11G11B10_FLOAT like the levels in the native-storage build -- Bloom.cpp:137)
    }
}
// The final up-sample with the chain's copy-frame pass as its tail (HnPostProcessTask.cpp:911-927: Bloom::Execute, then the draw that applies ToneMap() to
// the Bloom output): the texel goes to the Bloom output as before and, tone-mapped, to the LDR frame -- the arithmetic of tonemap_kernel on the value
// tonemap_kernel would have read back (this file is compiled without contraction, like tonemap.hip: bit-identical), one pass over the frame less.
template <bool STAGED, int MODE, bool SRGB>
__global__ __launch_bounds__(256) void bloom_final_tonemap_kernel(Img input, Img down, Img out, Img ldr, float intensity, float alphaInterp, ToneMapK tm, int writeOut, int srcPacked)
{
    __shared__ UpLds lds[STAGED ? kUpTW * kUpTH : 1];
    int x, y;
    v4  r;
    if (!bloom_upsample_texel<true, STAGED>(lds, input, down, out, intensity, alphaInterp, srcPacked, x, y, r)) return;
    if (writeOut) st<bloom_t>(out, x, y, r); // (0: nobody reads the Bloom output of this frame -- MIFX_CHAIN_FUSE_BLOOM_OUTPUT_ON_DEMAND; `out` still gives the rows)
    r = quantize_v4(r); // (the copy-frame pass reads the Bloom output as it is stored: a no-op in the fp32 build)
    v3 t = tone_map<MODE>(xyz(r), tm);
    if (SRGB) t = linear_to_srgb(t);
    st_v4_late<1>(ldr, x, y, mk4(t, r.w));
}

// ------------------------------------------------------------------------------------------------ the tail of the pyramid in one workgroup
// Below ~2000 texels a level is one or two workgroups of work and ~6 us of dispatch latency; the reference's default radius walks five such levels
// down and up again at 3840x2160 (60x33 ... 15x8).  One 1024-thread workgroup takes them all: level after level, a barrier in between, every texel
// computed by the code of the per-level kernels (the 13-tap sum on the un-staged source; the up-sample with the same staged / folded decision the
// per-level launcher would take, made on the host and handed over per level) -- bit-identical results, 2 x (levels - 1) dispatches less.
struct BloomTail
{
    Img      down[8], up[8]; // down[0] = the last level the wide kernels produced (source of the first tail level); up[i] pairs with down[i]
    int      levels;         // tail levels 1 .. levels - 1 are produced here
    unsigned foldMask;       // bit i: the up-sample that writes up[i] may take the folded 4x4 path (the launcher would have staged it)
};
// a tail level resident in LDS (w x h texels, row-major): the same fetch interface as Tile / Direct
struct LdsLevel
{
    static constexpr bool kZeroOutside = false;
    const v4* p;
    int       w, h;
    MIFX_D v4 fetch(int x, int y) const { return p[y * w + x]; }
};
template <class SRC> MIFX_D v3 bloom_upsample_sum(const SRC& src, int dw, int dh, int outW, int outH, int x, int y, bool mayFold)
{
    struct { int w, h; } down{dw, dh};
    const v2 uv = pixel_uv(x, y, outW, outH);
    const v2 ts{fdiv(1.0f, float(down.w)), fdiv(1.0f, float(down.h))};
    const TentAxis ax = tent_axis(uv.x, ts.x, down.w), ay = tent_axis(uv.y, ts.y, down.h);
    v3 sum;
    if (mayFold && ax.regular && ay.regular)
    {
        sum = mk3(0.0f);
#pragma unroll
        for (int j = 0; j < 4; ++j)
        {
            v3 row = mk3(0.0f);
#pragma unroll
            for (int i = 0; i < 4; ++i)
            {
                MIFX_FMA_BLOCK
                const v4 t = src.fetch(ax.i[i], ay.i[j]);
                row = v3{row.x + t.x * ax.w[i], row.y + t.y * ax.w[i], row.z + t.z * ax.w[i]};
            }
            {
                MIFX_FMA_BLOCK
                sum = v3{sum.x + row.x * ay.w[j], sum.y + row.y * ay.w[j], sum.z + row.z * ay.w[j]};
            }
            MIFX_TAP_FENCE(sum);
        }
    }
    else
    {
        auto S = [&](float ox, float oy) { v3 r = sample_linear_clamp_rgb(src, down.w, down.h, uv.x + ts.x * ox, uv.y + ts.y * oy); MIFX_TAP_FENCE(r); return r; };
        const v3 A = S(-1.0f, +1.0f), B = S(+0.0f, +1.0f), C = S(+1.0f, +1.0f);
        const v3 D = S(-1.0f, +0.0f), E = S(+0.0f, +0.0f), F = S(+1.0f, +0.0f);
        const v3 G = S(-1.0f, -1.0f), H = S(+0.0f, -1.0f), I = S(+1.0f, -1.0f);
        sum = E * 0.25f;
        sum += (B + D + F + H) * 0.125f;
        sum += (A + C + G + I) * 0.0625f;
    }
    return sum;
}
// The tail levels live in LDS while they are worked on (2048 + 512 + 128 + ... texels <= 44 KB): only the first level reads HBM (the last level of the wide
// kernels), every level is also written out (the per-level planes stay what mifx_bloom_get_intermediate and the wide up-sample of the next level read).
constexpr int kTailLdsTexels = 2048 + 2048 / 3 + 64; // (sized for up to 2048-texel first levels; mifx_bloom::kTailTexels decides which levels come here) // sum of the tail levels (each <= a quarter of the one before, odd sizes rounded down) with slack
__global__ __launch_bounds__(1024) void bloom_tail_kernel(BloomTail t)
{
    __shared__ v4 lds[kTailLdsTexels];
    __shared__ int off[9];
    const int tid = int(threadIdx.x), nthreads = int(blockDim.x);
    if (tid == 0)
    {
        int o = 0;
        for (int l = 1; l < t.levels; ++l) { off[l] = o; o += t.down[l].w * t.down[l].h; }
        off[t.levels] = o;
    }
    __syncthreads();
    for (int l = 1; l < t.levels; ++l) // B2 on the tail levels: level l from level l - 1 (HBM for the first, LDS after that)
    {
        const Img in = t.down[l - 1], out = t.down[l];
        v4* dst = lds + off[l];
        for (int i = tid; i < out.w * out.h; i += nthreads)
        {
            const int x = i % out.w, y = i / out.w;
            const v2  uv = pixel_uv(x, y, out.w, out.h);
            const Taps13 s = l == 1 ? fetch13(Direct<bloom_t>{in}, in.w, in.h, uv) : fetch13(LdsLevel{lds + off[l - 1], in.w, in.h}, in.w, in.h, uv);
            v3 c = mk3(0.0f);
            c += (s.A + s.C + s.G + s.I) * 0.03125f;
            c += (s.B + s.D + s.F + s.H) * 0.0625f;
            c += (s.E + s.J + s.K + s.L + s.M) * 0.125f;
            dst[i] = quantize_bloom(mk4(c, 0.0f)); // (the next level reads the level as it is stored)
            st<bloom_t>(out, x, y, mk4(c, 0.0f));
        }
        __syncthreads();
    }
    for (int l = t.levels - 1; l >= 2; --l) // B3: up[l - 1] = down[l - 1] + up-sample(l == last ? down[l] : up[l]); up[l - 1] replaces down[l - 1] in LDS
    {
        const Img srcImg = t.down[l], out = t.up[l - 1];
        const LdsLevel src{lds + off[l], srcImg.w, srcImg.h}; // holds down[l] for the last level, up[l] afterwards
        v4* acc = lds + off[l - 1];
        const bool mayFold = ((t.foldMask >> (l - 1)) & 1u) != 0u;
        for (int i = tid; i < out.w * out.h; i += nthreads)
        {
            const int x = i % out.w, y = i / out.w;
            const v3  sum = bloom_upsample_sum(src, src.w, src.h, out.w, out.h, x, y, mayFold);
            const v4  r   = mk4(xyz(acc[i]) + sum, 0.0f);
            acc[i] = quantize_bloom(r); // only this thread reads or writes texel i of this level in this step (the taps read level l)
            st<bloom_t>(out, x, y, r);
        }
        __syncthreads();
    }
}

static const dim3 kBlock(64, 4, 1);
static const dim3 kBloomBlock(kBX, kBY, 1);
static inline dim3 bloom_grid(const Img& out) { return dim3((out.w + kBX - 1) / kBX, (window_rows(out) + kBY - 1) / kBY, 1); }

mifx_status launch_bloom_prefilter(hipStream_t s, Img in, Img out, const mifx_bloom_attribs& a, bool packedInput)
{
    const bool staged = tile_fits(in.w, out.w, kBX, 2.0f, kDownTW) && tile_fits(in.h, out.h, kBY, 2.0f, kDownTH);
#define MIFX_B1(S, T) hipLaunchKernelGGL((bloom_prefilter_kernel<S, T>), bloom_grid(out), kBloomBlock, 0, s, in, out, a.Threshold, a.SoftTreshold)
    if (packedInput) { if (staged) MIFX_B1(true, bloom_t); else MIFX_B1(false, bloom_t); } // (fp32 build: bloom_t is v4 -- the same two kernels)
    else { if (staged) MIFX_B1(true, v4); else MIFX_B1(false, v4); }
#undef MIFX_B1
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_bloom_downsample(hipStream_t s, Img in, Img out)
{
    if (tile_fits(in.w, out.w, kBX, 2.0f, kDownTW) && tile_fits(in.h, out.h, kBY, 2.0f, kDownTH))
        hipLaunchKernelGGL((bloom_downsample_kernel<true>), bloom_grid(out), kBloomBlock, 0, s, in, out);
    else
        hipLaunchKernelGGL((bloom_downsample_kernel<false>), bloom_grid(out), kBloomBlock, 0, s, in, out);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_bloom_upsample(hipStream_t s, Img input, Img down, Img out, const mifx_bloom_attribs& a, bool final_pass, bool packedInput)
{
    const bool staged = tile_fits(down.w, out.w, kBX, 1.0f, kUpTW) && tile_fits(down.h, out.h, kBY, 1.0f, kUpTH);
#define MIFX_UP(F, S) hipLaunchKernelGGL((bloom_upsample_kernel<F, S>), bloom_grid(out), kBloomBlock, 0, s, input, down, out, a.Intensity, a.AlphaInterpolation, packedInput ? 1 : 0)
    if (final_pass) { if (staged) MIFX_UP(true, true); else MIFX_UP(true, false); }
    else { if (staged) MIFX_UP(false, true); else MIFX_UP(false, false); }
#undef MIFX_UP
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
// Tail of the pyramid: down[first .. last] from down[first - 1], then up[last - 1 .. first] (up[first - 1] and the levels above stay with the wide kernels).
// `down` / `up`: views of the levels first - 1 .. last (index 0 = level first - 1).
bool bloom_tail_fits(const Img* down, int count) // the tail levels (index 1 .. count - 1) fit the kernel's LDS
{
    int texels = 0;
    for (int i = 1; i < count; ++i) texels += down[i].w * down[i].h;
    return count >= 2 && count <= 8 && texels <= kTailLdsTexels;
}
mifx_status launch_bloom_tail(hipStream_t s, const Img* down, const Img* up, int count)
{
    if (count < 2 || count > 8) { set_error("launch_bloom_tail: %d levels", count); return MIFX_ERR_INVALID_ARG; }
    {
        int texels = 0;
        for (int i = 1; i < count; ++i) texels += down[i].w * down[i].h;
        if (texels > kTailLdsTexels) { set_error("launch_bloom_tail: %d texels exceed the LDS budget of %d", texels, kTailLdsTexels); return MIFX_ERR_INVALID_ARG; }
    }
    BloomTail t{};
    t.levels = count;
    for (int i = 0; i < count; ++i) { t.down[i] = down[i]; t.up[i] = up[i]; }
    for (int l = count - 1; l >= 2; --l) // the decision launch_bloom_upsample takes for the same pair of sizes
    {
        const Img& src = t.down[l]; // (down[l] and up[l] have the same size)
        const Img& out = t.up[l - 1];
        if (tile_fits(src.w, out.w, kBX, 1.0f, kUpTW) && tile_fits(src.h, out.h, kBY, 1.0f, kUpTH)) t.foldMask |= 1u << (l - 1);
    }
    hipLaunchKernelGGL(bloom_tail_kernel, dim3(1, 1, 1), dim3(1024, 1, 1), 0, s, t);
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
mifx_status launch_bloom_final_tonemap(hipStream_t s, Img input, Img down, Img out, Img ldr, const mifx_bloom_attribs& a, const mifx_tone_mapping_attribs& attr, float ave_log_lum,
                                       uint32_t flags, bool writeBloomOutput, bool packedInput)
{
    const bool staged = tile_fits(down.w, out.w, kBX, 1.0f, kUpTW) && tile_fits(down.h, out.h, kBY, 1.0f, kUpTH);
    const bool srgb   = (flags & MIFX_TONEMAP_FLAG_CONVERT_OUTPUT_TO_SRGB) != 0;
    const ToneMapK tm = make_tonemapk(attr, ave_log_lum);
#define MIFX_FT(S, M, G) hipLaunchKernelGGL((bloom_final_tonemap_kernel<S, M, G>), bloom_grid(out), kBloomBlock, 0, s, input, down, out, ldr, a.Intensity, a.AlphaInterpolation, tm, writeBloomOutput ? 1 : 0, packedInput ? 1 : 0)
#define MIFX_FT_MODE(M)                                                     \
    if (staged) { if (srgb) MIFX_FT(true, M, true); else MIFX_FT(true, M, false); }    \
    else { if (srgb) MIFX_FT(false, M, true); else MIFX_FT(false, M, false); }
    MIFX_TONEMAP_DISPATCH(attr.iToneMappingMode, MIFX_FT_MODE)
#undef MIFX_FT_MODE
#undef MIFX_FT
    MIFX_HIP_CHECK(hipGetLastError());
    return MIFX_OK;
}
} // namespace mifx
