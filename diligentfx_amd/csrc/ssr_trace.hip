// MIFX_BUILD_FLAGS: -mllvm -amdgpu-sched-strategy=max-memory-clause
// (build.py reads the line above.  The backend's memory-clause scheduling strategy for this file only: R4 342 -> 330 us in two A/B runs, bit-identical output; the other
//  strategies and the other files lose or do not move -- max-ilp: TAA +19 %, Bloom's final pass +32 %; max-memory-clause on ssr_temporal.hip: R6 +12 % --
//  profiles/r04_ab_sched_strategy.txt.)
// ssr_trace.hip -- ScreenSpaceReflection pass R4 (ray generation + hierarchical march; the other passes are in ssr.hip) (AMD-SSSR-derived stochastic screen-space reflections).
// Math follows Shaders/PostProcess/ScreenSpaceReflection/private/SSR_*.fx; host sequence in api_ssr.cpp.
//
// Masking: the reference marks reflection samples in a D16 depth target and depth-tests R4-R7 against it
// (ScreenSpaceReflection.cpp:47-51,550,626,...).  Here the mask is a float plane (1 = reflection sample); R4 writes 0 to masked-out texels of its two targets, which the
// reference clears every frame (R5 / R6 leave theirs alone, as the reference's depth test does: ssr.hip).
#include "mifx_host.h"
#include "mifx_effects.h"
#include "mifx_pbr.h"
#include <cstdlib>

namespace mifx
{


// ------------------------------------------------------------------------------------------------ R4: intersection (SSR_ComputeIntersection.fx:31-335)
// Texture.Load on the depth hierarchy: out of bounds -> 0.  All levels live in one allocation (HizSlab), so a tap is one 32-bit offset from a
// uniform base: the level record {offset, pitch, w, h} comes from LDS (one ds_read_b128), the address is a 32-bit multiply-add.  The slab is read
// through a buffer resource (buffer_load_dword ... offen): an offset computed from coordinates outside the level can be anything, the range check of
// the descriptor turns it into a harmless read (0 beyond the slab), and the value is discarded by the bounds test anyway -- no address select per tap.
// One LDS record per level: the address record and {MipResolution, rcp(MipResolution)} side by side, so that a march step fetches both with one address as soon
// as the next level is known (two ds_read_b128 in flight together instead of one at the end of a step and a dependent one at the start of the next).
struct HizLevel
{
    uint4 addr; // {offset, pitch, w, h}
    v4    res;  // {w_f, h_f, 1 / w_f, 1 / h_f} of the level as the reference carries them
};
// The table has one entry in front of level 0 (a copy of it): the march keeps its level as the byte offset of the entry, (level + 1) * sizeof(HizLevel), and a
// ray that leaves the most detailed level (level -1 when that is level 0) still reads an entry.
struct HizLds
{
    __amdgpu_buffer_rsrc_t rsrc;
    const HizLevel*        lv;
    __amdgpu_buffer_rsrc_t rsrc0; // DIRECT0: level 0 = the caller's depth plane, read where it lies (HizSlab::base0)
};
// DIRECT0 (round 6): level 0 is not in the slab.  A tap picks its descriptor by the lane's level -- two loads under complementary lane masks, one v_cmp and a few scalar
// instructions more per tap; the offset arithmetic is the same (the level-0 record holds offset 0 and the plane's own pitch).  `level0`: this lane's tap is a level-0 tap.
template <bool DIRECT0> MIFX_D float load_hiz(const HizLds& hz, uint4 L, int x, int y, bool level0)
{
    const bool     in  = unsigned(x) < L.z && unsigned(y) < L.w;
    // offset + y * pitch + x * 4 in two instructions (the compiler's own choice is a multiply, a shift and a three-operand add); the 24-bit multiply is exact for
    // every row inside the level (pitch, rows < 2^24)
    unsigned row, off;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(y), "v"(L.y), "v"(L.x));
    asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(x), "v"(row));
    float v;
    if (DIRECT0)
    {
        // two loads under complementary lane masks.  (Written plainly, the compiler merges the two calls into one load whose descriptor is a per-lane select and wraps it in
        // a readfirstlane loop -- twice the scalar work and a serialised pair of loads; the differing empty asm statements keep the branches apart.)
        if (level0)
        {
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hz.rsrc0, int(off), 0, 0));
            asm volatile("; level 0: the depth plane" : : "v"(off)); // (an operand that is ready: nothing waits for the load here)
        }
        else
        {
            v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hz.rsrc, int(off), 0, 0));
            asm volatile("; levels 1 ..: the slab" : : "v"(off));
        }
    }
    else v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hz.rsrc, int(off), 0, 0));
    return in ? v : 0.0f;
}
template <bool DIRECT0> MIFX_D float load_hiz(const HizLds& hz, int x, int y, int mip) { return load_hiz<DIRECT0>(hz, hz.lv[mip + 1].addr, x, y, mip == 0); }

// HizLevel::res of level m = {MipResolution, rcp(MipResolution)}.  The reference carries both through the loop with exact *2 / *0.5 updates
// (:176-178), so they only ever take the values screen * 2^-m and 1 / (screen * 2^-m): the per-level table in LDS returns the identical
// floats and takes six vector instructions and a branch out of every march step.
// The multiply-adds of a march step stay separate multiplies and adds, as the reference's shader compiler emits them for an fp32 target without contraction.
// (-DMIFX_R4_FUSED_MARCH fuses 8 of the step's 45 vector instructions: kernel -3 % in round 2, but 1.4e-3 of the specular values of a 224 x 96 frame then differ by
// more than 1e-3 -- rays that cross a tile edge at another step -- where the separate form has none: profiles/r04_parity_outliers_strict_vs_fast.txt.  Round 4 made the
// exact form the build: parity first, see build.py.)
// Round 3, measured and rejected on the MI355X (tools/ab_gpu.sh; profiles/r03_ab_mlp.txt, r03_ab_r4_alignment.txt, r03_ab_r4_epilogue.txt; every variant passed the
// parity suite).  The counters say the kernel is latency-bound (74 % of the wave cycles parked on s_waitcnt, 13 % issuing VALU, eight waves per SIMD), yet:
//   * the records of both candidate next levels requested from LDS beside the depth tap and selected afterwards (the LDS round trip out of the step's dependent
//     chain): 341 -> 363 us (re-measured over 60 frames at the steady-state clock: 310 -> 330 us) -- eight selects and two more LDS reads per step cost more than the
//     latency they hide;
//   * the pdf, the hit confidence and the edge vignette on 1-ulp reciprocals / square roots with the per-frame quotients (2 / screen, 0.005 2^mip / screen, 1 / fov,
//     1 / thickness) computed on the host: 74 vector instructions less per ray, 328 -> 343 us with an instruction-for-instruction identical march loop;
//   * the march loop aligned to 64 bytes (.p2align, with and without a 32-byte offset): 343 / 342 / 341 us, nothing;
//   * the three loads of the hit validation (depth hierarchy, normal, colour at the hit) issued as one group right after the march instead of behind its early
//     exits: 330 -> 333 us.
//   * the hierarchy stored as 8 x 4-texel tiles of one 128-byte line each (a twin written by the pyramid kernels; six address instructions per tap instead of two),
//     because the L1 counters show 55 of the 64 lanes of a depth tap on a line of their own (profiles/r03_pmc_tcp1_v8.txt): 328.9 -> 330.4 us
//     (profiles/r03_ab_hiz_tiled.txt; re-measured over 60 frames at the steady-state clock: 313.1 vs 313.2 us) -- the tag lookups are not the limit either;
//   * the tiles handed to the XCDs in 128 x 32-pixel chunks so that an L2 serves neighbouring tiles (mifx_device.h, tiled_xy): +3.5 %.
//   * the two coarsest levels of the hierarchy (120x67 + 60x33 texels at 4K, 40 KB) resident in LDS, because HALF of all march steps run there (49 % at levels >= 5,
//     62 % at >= 4: tools/r4_stats.py), as 512-thread workgroups that fill their copy per 512 pixels: 311 -> 418 us (the fill is 80 bytes per pixel); as persistent
//     waves (the grid fills the device once, every workgroup fills its cache once, every wave walks its own sequence of tiles): 400 us, against 418 us for the same
//     persistent kernel WITHOUT the cache -- the cache is worth 4 %, the static tile sequence costs 34 % (profiles/r03_ab_r4_lds_cache.txt).  The taps at the coarse
//     levels were L1 hits already; the step time is set by the taps at the fine levels, which the cache does not touch.  All variants bit-identical.
// What is left is the march itself: 47 steps per wave on average (38.6 per ray; the lanes of a wave are 80 % busy, profiles/r03_r4_march_steps.txt), each a chain
// of a dependent L1/L2 round trip, an LDS read and ~32 vector instructions, at 5.3 resident waves per SIMD on average: ~480 ns per step, 46 % of it covered by the
// other waves' arithmetic.
// REV = SSR_OPTION_INVERTED_DEPTH (:108-113, 118-124): larger depth is closer to the camera
template <bool REV, bool DIRECT0>
MIFX_D v3 hierarchical_raymarch(const HizLds& hiz, v3 origin, v3 dir, v2 screen, int mostDetailedMip, unsigned maxIter, bool& validHit, unsigned& steps) // :139-189
{
    const v3 invDir{dir.x != 0.0f ? fdiv(1.0f, dir.x) : SSR_FLT_MAX, dir.y != 0.0f ? fdiv(1.0f, dir.y) : SSR_FLT_MAX, dir.z != 0.0f ? fdiv(1.0f, dir.z) : SSR_FLT_MAX};
    constexpr int kEntry = int(sizeof(HizLevel));
    const int     loMin  = (mostDetailedMip + 1) * kEntry;
    int           lo     = loMin; // (CurrentMip + 1) * sizeof(HizLevel)
    // (two 16-byte vector loads: a struct copy is split into scalar LDS reads that end up at the top of the step, in front of the address arithmetic)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto entry = [&](int o) {
        const char* p = reinterpret_cast<const char*>(hiz.lv) + o;
        const u32x4   a = *reinterpret_cast<const u32x4*>(p);
        const mifx_f4 r = *reinterpret_cast<const mifx_f4*>(p + 16);
        return HizLevel{uint4{a.x, a.y, a.z, a.w}, v4{r.x, r.y, r.z, r.w}};
    };
    HizLevel L = entry(lo);
    v2  mipRes{L.res.x, L.res.y};
    v2  invMipRes{L.res.z, L.res.w};
    // t.z of a step is (surfaceDepth - origin.z) / dir.z for a ray that moves away from the camera and FLT_MAX otherwise (:118-124).  The choice is per ray, so it
    // is folded into the two constants of the multiply-add: depth * 0 + FLT_MAX is FLT_MAX exactly for every finite depth.
    const bool  away = REV ? dir.z < 0.0f : dir.z > 0.0f;
    const float tzMul = away ? invDir.z : 0.0f, tzAdd = away ? -(origin.z * invDir.z) : SSR_FLT_MAX;
    v2  uvOffset = (0.005f * float(1 << mostDetailedMip)) / screen;
    uvOffset.x = dir.x < 0.0f ? -uvOffset.x : uvOffset.x;
    uvOffset.y = dir.y < 0.0f ? -uvOffset.y : uvOffset.y;
    const v2 floorOffset{dir.x < 0.0f ? 0.0f : 1.0f, dir.y < 0.0f ? 0.0f : 1.0f};

    // InitialAdvanceRay :66-86
    float curT;
    v3    pos;
    {
        const v2 mp = mipRes * mk2(origin.x, origin.y);
        v2 plane{floorf(mp.x) + floorOffset.x, floorf(mp.y) + floorOffset.y};
        plane = plane * invMipRes + uvOffset;
        const v2 t{plane.x * invDir.x - origin.x * invDir.x, plane.y * invDir.y - origin.y * invDir.y};
        curT = fminf(t.x, t.y);
        pos  = origin + curT * dir;
    }
    unsigned idx = 0u;
#ifdef MIFX_R4_STATS
    unsigned coarse5 = 0u, coarse4 = 0u; // steps taken at hierarchy levels >= 5 / >= 4 (tools/r4_stats.py)
#endif
    // Round 6: the march runs at raised wave priority.  A SIMD's eight waves are in different phases -- some in the ~960 vector instructions of the ray set-up or in the
    // hit validation, some in the march, where every step is a dependent chain (tap -> compare -> next level -> LDS record -> next tap) of ~30 instructions.  At equal
    // priority a marching wave whose tap has arrived queues behind the set-up arithmetic of its neighbours; in front of them it asks for its next tap sooner, and the
    // set-up waves lose nothing they were not going to wait for anyway.  s_setprio 3 around the loop: 307.8 / 306.6 / 308.6 / 309.4 -> 301.7 / 301.6 / 303.1 / 304.6 us in
    // four A/B runs on three boxes (-1.7 %; priority 2 the same, 1 half of it; kept through the hit validation as well: 305.9); same instructions, same values
    // (profiles/r06_ab_r4_setprio.txt).  Raised priority from a wave's start until its first loads are out as well: 304.5 / 304.6 us, nothing.  The same idea for the load groups of the straight-line kernels (R6, TAA: raised priority until a wave's loads are out) loses 1 - 3 %.
#ifndef MIFX_R4_PRIO
#define MIFX_R4_PRIO 3
#endif
#if MIFX_R4_PRIO
    __builtin_amdgcn_s_setprio(MIFX_R4_PRIO);
#endif
    while (idx < maxIter && lo >= loMin)
    {
#ifdef MIFX_R4_STATS
        coarse5 += lo >= 6 * kEntry ? 1u : 0u;
        coarse4 += lo >= 5 * kEntry ? 1u : 0u;
#endif
        const v2    mp = mipRes * mk2(pos.x, pos.y);
        const float surfaceDepth = load_hiz<DIRECT0>(hiz, L.addr, int(mp.x), int(mp.y), lo <= kEntry);
        // AdvanceRay :88-137
        v2 plane{floorf(mp.x) + floorOffset.x, floorf(mp.y) + floorOffset.y};
#ifdef MIFX_R4_FUSED_MARCH
        plane = v2{__builtin_fmaf(plane.x, invMipRes.x, uvOffset.x), __builtin_fmaf(plane.y, invMipRes.y, uvOffset.y)};
        v3 t{__builtin_fmaf(plane.x, invDir.x, -(origin.x * invDir.x)), __builtin_fmaf(plane.y, invDir.y, -(origin.y * invDir.y)), __builtin_fmaf(surfaceDepth, tzMul, tzAdd)};
#else
        plane = plane * invMipRes + uvOffset;
        v3 t{plane.x * invDir.x - origin.x * invDir.x, plane.y * invDir.y - origin.y * invDir.y, surfaceDepth * invDir.z - origin.z * invDir.z};
        t.z = away ? t.z : SSR_FLT_MAX;
#endif
        const float tmin = fminf(fminf(t.x, t.y), t.z);
        const bool  above = REV ? surfaceDepth < pos.z : surfaceDepth > pos.z;
        const bool  skipped = __float_as_uint(tmin) != __float_as_uint(t.z) && above;
        curT = above ? tmin : curT;
#ifdef MIFX_R4_FUSED_MARCH
        pos = v3{__builtin_fmaf(curT, dir.x, origin.x), __builtin_fmaf(curT, dir.y, origin.y), __builtin_fmaf(curT, dir.z, origin.z)};
#else
        pos  = origin + curT * dir;
#endif

        // CurrentMip += SkippedTile ? 1 : -1 unless that would leave the generated levels (:171-179); CurrentMip never exceeds SSR_MAX_MIP (MostDetailedMip is
        // validated against it), so "stay" is the upper clamp
        lo = min(lo + (skipped ? kEntry : -kEntry), (SSR_MAX_MIP + 1) * kEntry);
        L  = entry(lo);
        mipRes    = v2{L.res.x, L.res.y};
        invMipRes = v2{L.res.z, L.res.w};
        ++idx;
    }
#if MIFX_R4_PRIO
    __builtin_amdgcn_s_setprio(0);
#endif
    validHit = true; // ValidHit = (i <= MaxTraversalIntersections) :187 -- the loop cannot leave i above the bound
    steps = idx;
#ifdef MIFX_R4_STATS
    steps = idx + 256u * coarse5 + 65536u * coarse4;
#endif
    return pos;
}
MIFX_D float smoothstepf(float a, float b, float x)
{
    const float t = saturate(fdiv(x - a, b - a));
    return t * t * (3.0f - 2.0f * t);
}
MIFX_D float edge_vignette(v2 hit, v2 screen) // CalculateEdgeVignette :191-196
{
    const v2 fov{0.05f * fdiv(screen.y, screen.x), 0.05f * 1.0f};
    const v2 border{smoothstepf(0.0f, fov.x, hit.x) * (1.0f - smoothstepf(1.0f - fov.x, 1.0f, hit.x)),
                    smoothstepf(0.0f, fov.y, hit.y) * (1.0f - smoothstepf(1.0f - fov.y, 1.0f, hit.y))};
    return border.x * border.y;
}
// hitPrev (PREV only): the hit moved back along its motion vector, SSR_OPTION_PREVIOUS_FRAME :230-231
template <bool PREV, bool REV, bool DIRECT0>
MIFX_D float validate_hit(const HizLds& hiz, const Img& normalTex, v3 hit, v2 hitPrev, v2 uv, v3 rayDirWS, v2 screen, float thickness, const m44& proj) // ValidateHit :199-252
{
    if (hit.x < 0.0f || hit.y < 0.0f || hit.x > 1.0f || hit.y > 1.0f) return 0.0f;
    const v2 manhattan{fabsf(hit.x - uv.x), fabsf(hit.y - uv.y)};
    if (manhattan.x < fdiv(2.0f, screen.x) && manhattan.y < fdiv(2.0f, screen.y)) return 0.0f;
    const int   tx = int(screen.x * hit.x), ty = int(screen.y * hit.y);
    const float surfaceDepth = load_hiz<DIRECT0>(hiz, tx, ty, 0);
    if (is_background(surfaceDepth, REV)) return 0.0f;
    const v3 hitNormal = (tx < 0 || ty < 0 || tx >= normalTex.w || ty >= normalTex.h) ? mk3(0.0f) : xyz(ld<v4>(normalTex, tx, ty));
    if (dot(hitNormal, rayDirWS) > 0.0f) return 0.0f;
    const v3    surfaceVS = screen_xy_depth_to_view_space(v3{hit.x, hit.y, surfaceDepth}, proj);
    const v3    hitVS     = screen_xy_depth_to_view_space(hit, proj);
    const float dist      = length(surfaceVS - hitVS);
    const float vignette  = PREV ? fminf(edge_vignette(hitPrev, screen), edge_vignette(mk2(hit.x, hit.y), screen)) : edge_vignette(mk2(hit.x, hit.y), screen);
    float confidence = 1.0f - smoothstepf(0.0f, thickness, dist * fdiv(1.0f, surfaceVS.z + SSR_FLT_EPS));
    confidence *= confidence;
    return vignette * confidence;
}

// PREV = FEATURE_FLAG_PREVIOUS_FRAME: `radiance` is last frame's colour; the hit is reprojected with the motion vector at the hit (:310-314)
template <bool PREV, bool REV, bool DIRECT0 = false>
#ifndef MIFX_R4_WAVES
#define MIFX_R4_WAVES 0
#endif
#ifdef MIFX_R4_CAP // experiment knob: at most this many waves per SIMD (leaves wave slots to a kernel that runs beside the march on another stream)
#define MIFX_R4_OCC __attribute__((amdgpu_waves_per_eu(MIFX_R4_CAP, MIFX_R4_CAP)))
#else
#define MIFX_R4_OCC MIFX_WAVES_OPT(MIFX_R4_WAVES)
#endif
// Scalar registers: a CU admits min(8, 800 / (16 ceil(sgprs / 16) + 16)) workgroups of 256 threads (MI355X_MICROARCH.md, "Residency"): with the 84 - 88 the allocator takes
// when left alone that is 7, with <= 80 it is 8 -- and the march's time follows the resident waves (6 workgroups per CU: +10 %, 4: +37 %, profiles/r05_ab_occupancy_r4_a3.txt).
#ifndef MIFX_R4_SGPRS
#define MIFX_R4_SGPRS 72
#endif
#if MIFX_R4_SGPRS > 0
#define MIFX_R4_SGPR_CAP __attribute__((amdgpu_num_sgpr(MIFX_R4_SGPRS)))
#else
#define MIFX_R4_SGPR_CAP
#endif
__global__ __launch_bounds__(256) MIFX_R4_OCC MIFX_R4_SGPR_CAP void ssr_intersection_kernel(Img radiance, Img normalTex, Img roughnessTex, Img noiseXY, HizSlab hizSlab, Img mask, Img motionTex, Img outSpec,
                                                               Img outDirPdf, CamK cam, SsrK k, Img hitCoords, int localBegin, int localEnd)
{
    __shared__ HizLevel hizLv[SSR_MAX_MIP + 2];
    // (round 5: the texel's reflection mask is requested before the level table is staged -- the table's rows come out of the kernel arguments by a vector load and are
    //  followed by a barrier, a round trip the mask's now shares; clamped coordinates for the threads outside the image, which drop the value)
    int x, y;
    const bool  inImage   = tiled_xy_xcd(outSpec, x, y);
    const float maskValue = ld<mask_t>(mask, min(x, outSpec.w - 1), min(y, row_end(outSpec) - 1));
    if (threadIdx.x < unsigned(SSR_MAX_MIP + 2))
    {
        const unsigned m = threadIdx.x == 0u ? 0u : threadIdx.x - 1u; // entry 0 = a second copy of level 0
        const float s = fdiv(1.0f, float(1 << int(m)));
        const v2    r{cam.vw * s, cam.vh * s};
        hizLv[threadIdx.x] = HizLevel{uint4{hizSlab.offset[m], hizSlab.pitch[m], hizSlab.w[m], hizSlab.h[m]}, v4{r.x, r.y, fdiv(1.0f, r.x), fdiv(1.0f, r.y)}};
    }
    __syncthreads();
    const HizLds hiz{__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hizSlab.base), 0, int(hizSlab.bytes), 0x00020000), hizLv,
                     __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(DIRECT0 ? hizSlab.base0 : hizSlab.base), 0, int(DIRECT0 ? hizSlab.bytes0 : hizSlab.bytes), 0x00020000)};
    if (!inImage) return;
    if (maskValue == 0.0f)
    {
        st<v4>(outSpec, x, y, mk4(0.0f)); // both targets are cleared to 0 (ScreenSpaceReflection.cpp:993-994)
        st<v4>(outDirPdf, x, y, mk4(0.0f));
        if (hitCoords.p != nullptr) st<float>(hitCoords, x, y, __uint_as_float(0xffffffffu));
        return;
    }
    const v2 screen{cam.vw, cam.vh};
    // the full-resolution pixel whose ray this texel traces: itself, or (half resolution, :283-288) one pixel of its 2x2 block chosen by
    // ComputeHalfResolutionOffset (PostFX_Common.fxh:45-55); targets, mask and noise stay indexed by the texel (x, y)
    int px = x, py = y;
    if (k.HalfResolution)
    {
        const unsigned sampleIdx = (1320229860u >> (((unsigned(x) & 3u) << 3u) + ((unsigned(y) & 3u) << 1u))) & 3u;
        px = 2 * x + int(sampleIdx & 1u);
        py = 2 * y + int(sampleIdx >> 1u);
    }
    const v2 uv{(float(px) + 0.5f) * cam.ivw, (float(py) + 0.5f) * cam.ivh};
    const v3 normalVS  = mul_dir(xyz(ld<v4>(normalTex, px, py)), cam.view);
    const float rough  = ld<rough_t>(roughnessTex, px, py);
    const bool mirror  = rough < 0.01f; // IsMirrorReflection
    const int  mdm     = mirror ? 0 : int(k.MostDetailedMip);
    const v2   mipRes  = screen * fdiv(1.0f, float(1 << mdm));
    const v3   originSS{uv.x, uv.y, load_hiz<DIRECT0>(hiz, int(uv.x * mipRes.x), int(uv.y * mipRes.y), mdm)};
    const v3   originVS = screen_xy_depth_to_view_space(originSS, cam.proj);

    // SampleReflectionVector :254-278 (GGX VNDF, spherical caps)
    const v3 view = -normalize(originVS);
    v3 dirVS;
    float pdf;
    {
        const float alpha = rough * rough;
        const v3 N = normalVS;
        const v3 T = normalize(cross(N, fabsf(N.y) > 0.5f ? v3{1.0f, 0.0f, 0.0f} : v3{0.0f, 1.0f, 0.0f}));
        const v3 B = cross(T, N);
        v2 xi = ld<v2>(noiseXY, x & 127, y & 127);
        xi.y  = lerpf(xi.y, 0.0f, k.GGXImportanceSampleBias);
        const v3 viewTS{dot(T, view), dot(B, view), dot(N, view)};
        const v3 micro  = smith_ggx_sample_visible_normal_sc(viewTS, alpha, alpha, xi.x, xi.y);
        const v3 sampTS = reflect(-viewTS, micro);
        const float NdotV = viewTS.z, NdotH = micro.z;
        const float D  = normal_distribution_ggx(NdotH, alpha);
        const float G1 = smith_ggx_masking(NdotV, alpha);
        pdf   = fdiv(G1 * D, 4.0f * NdotV + SSR_FLT_EPS);
        dirVS = sampTS.x * T + sampTS.y * B + sampTS.z * N;
    }
    const v3 dirSS = project_position(originVS + dirVS, cam.proj) - originSS; // ProjectDirection
    const v3 dirWS = mul_dir(dirVS, cam.viewInv);

    bool validHit = false;
    unsigned steps = 0u;
    const v3 hitSS = hierarchical_raymarch<REV, DIRECT0>(hiz, originSS, dirSS, screen, mdm, k.MaxTraversalIntersections, validHit, steps);
#ifdef MIFX_R4_STATS // tools/r4_stats.py: the number of march steps of every ray in place of the pdf
    pdf = float(steps);
#endif
    const v3 hitVS = screen_xy_depth_to_view_space(hitSS, cam.proj);
    v2 hitPrev{hitSS.x, hitSS.y};
    if (PREV && validHit)
    {
        const v2 m = ld_zero_v2(motionTex, int(screen.x * hitSS.x), int(screen.y * hitSS.y)); // LoadMotion :56-59
        hitPrev = v2{hitSS.x - m.x * 0.5f, hitSS.y - m.y * -0.5f};
    }
    const float confidence = validHit ? validate_hit<PREV, REV, DIRECT0>(hiz, normalTex, hitSS, hitPrev, uv, dirWS, screen, k.DepthBufferThickness, cam.proj) : 0.0f;
    v3 refl = mk3(0.0f);
    // Row-band sharding (hitCoords.p != null, uniform): the colour at the hit may lie in a row this rank did not shade.  For such a hit the march only records WHERE it
    // is (0xffffffff: nothing to fetch) and pbr_hit_fetch_kernel (pbr.hip) shades that pixel on the spot; a hit in the rows [localBegin, localEnd) this rank shaded itself
    // is loaded here as in the unsharded frame (round 5: until then the fetch pass loaded those too -- 52 B per ray texel through a second kernel, 71 us per band at 8K).
    unsigned where = 0xffffffffu;
    if (confidence > 0.0f)
    {
        const int  rx = int(screen.x * hitPrev.x), ry = int(screen.y * hitPrev.y);
        const bool in = rx >= 0 && ry >= 0 && rx < radiance.w && ry < radiance.h;
        if (hitCoords.p != nullptr && (ry < localBegin || ry >= localEnd)) where = in ? unsigned(rx) | (unsigned(ry) << 16) : 0xffffffffu;
        else if (in) refl = xyz(ld<v4>(radiance, rx, ry));
    }
    if (hitCoords.p != nullptr) st<float>(hitCoords, x, y, __uint_as_float(where));
    st<v4>(outSpec, x, y, mk4(refl, confidence));
    st<v4>(outDirPdf, x, y, mk4(dirWS * length(hitVS - originVS), pdf));
}


// ------------------------------------------------------------------------------------------------ R4 with two rays per lane (round 5; -DMIFX_R4_TWO_RAYS=1, not in the shipped build)
// MEASURED AND NOT TAKEN (profiles/r05_ab_r4_two_rays.txt): 320.5 us against 310.1 us for the one-ray kernel at eight workgroups per CU (the SGPR cap above), same box,
// bit-identical output (tools/variant_hash.py at two sizes, tests/test_gpu_ssr.py green).  Fourteen rays per SIMD in flight instead of eight buy nothing: a step is as
// long as its slowest lane's tap, and 128 rays per wave have a slower slowest lane than 64 -- the march is bound by the rate at which the L1 hands out the taps of
// rays that share no cache line, not by how many are waiting.  Kept behind the macro so that the measurement can be repeated (tools/make_variant.py r4x2 -DMIFX_R4_TWO_RAYS=1).
#ifndef MIFX_R4_TWO_RAYS
#define MIFX_R4_TWO_RAYS 0
#endif
#if MIFX_R4_TWO_RAYS
// The march is a chain of dependent loads: one tap of the depth hierarchy per step, each waiting for the slowest of a wave's 64 rays (4 of which miss the L1 on average,
// so that nearly every step pays an L2 or Infinity-Cache round trip), and a SIMD holds at most eight waves whatever the register count.  The kernel's time follows the
// rays in flight (profiles/r05_ab_occupancy_r4_a3.txt: 6 workgroups per CU instead of 7 +10 %, 4 +37 %), and the one-ray kernel needs only 50 registers.  Here a lane
// carries TWO rays -- a wave covers a 16 x 8 pixel block, lane l the pixels l of its left and right 8 x 8 tile -- and a march step issues both taps before it uses
// either: six waves per SIMD x 2 rays = 12 rays per SIMD in flight instead of 8.  Per ray the arithmetic is the one-ray kernel's, statement for statement (same
// helpers, same operation order: bit-identical output, tests/test_gpu_ssr.py compares the two).  What the tail of a ray needs after the march (uv, world-space
// direction, view-space origin, pdf) waits in LDS meanwhile, so that the loop holds two march states and nothing else.
struct R4March // the march state of one ray: AdvanceRay's loop-invariant inputs and what a step updates
{
    v3    origin, dir, invDir;
    v2    uvOffset, floorOffset;
    bool  away;  // the ray moves away from the camera: t.z is the intersection with the surface depth, FLT_MAX otherwise (:118-124)
    int   lo, loMin; // (CurrentMip + 1) * sizeof(HizLevel); lo < loMin: the ray has left the most detailed level -- or there is no ray (masked-out texel)
    float curT;
    v3    pos;
};
struct R4Tail // per ray, parked in LDS during the march
{
    v2    uv;
    v3    dirWS, originVS;
    float pdf;
};
constexpr int kR4TailFloats = 9;
MIFX_D void r4_park(float* slot, const R4Tail& t) // slot = &lds[ray][0][thread]; fields 256 floats apart (one bank per lane)
{
    slot[0 * 256] = t.uv.x; slot[1 * 256] = t.uv.y; slot[2 * 256] = t.dirWS.x; slot[3 * 256] = t.dirWS.y; slot[4 * 256] = t.dirWS.z;
    slot[5 * 256] = t.originVS.x; slot[6 * 256] = t.originVS.y; slot[7 * 256] = t.originVS.z; slot[8 * 256] = t.pdf;
}
MIFX_D R4Tail r4_unpark(const float* slot)
{
    return R4Tail{v2{slot[0 * 256], slot[1 * 256]}, v3{slot[2 * 256], slot[3 * 256], slot[4 * 256]}, v3{slot[5 * 256], slot[6 * 256], slot[7 * 256]}, slot[8 * 256]};
}

// Everything of the one-ray kernel up to the march, for the texel (x, y).  false: no ray (outside the image / the row window, or masked out -- the targets then got
// their cleared values here), and m.lo < m.loMin so that the march leaves the slot alone.
template <bool REV>
MIFX_D bool r4_setup(const HizLds& hiz, const Img& normalTex, const Img& roughnessTex, const Img& noiseXY, const Img& mask, const Img& outSpec, const Img& outDirPdf, const Img& hitCoords,
                     const CamK& cam, const SsrK& k, int x, int y, R4March& m, R4Tail& tail)
{
    constexpr int kEntry = int(sizeof(HizLevel));
    m.loMin = kEntry;
    m.lo    = 0; // no ray
    m.origin = m.dir = m.invDir = m.pos = mk3(0.0f);
    m.uvOffset = m.floorOffset = mk2(0.0f, 0.0f);
    m.away = false;
    m.curT = 0.0f;
    tail   = R4Tail{mk2(0.0f, 0.0f), mk3(0.0f), mk3(0.0f), 0.0f};
    if (!(x < outSpec.w && y < row_end(outSpec))) return false;
    if (ld<mask_t>(mask, x, y) == 0.0f)
    {
        st<v4>(outSpec, x, y, mk4(0.0f)); // both targets are cleared to 0 (ScreenSpaceReflection.cpp:993-994)
        st<v4>(outDirPdf, x, y, mk4(0.0f));
        if (hitCoords.p != nullptr) st<float>(hitCoords, x, y, __uint_as_float(0xffffffffu));
        return false;
    }
    const v2 screen{cam.vw, cam.vh};
    int px = x, py = y;
    if (k.HalfResolution)
    {
        const unsigned sampleIdx = (1320229860u >> (((unsigned(x) & 3u) << 3u) + ((unsigned(y) & 3u) << 1u))) & 3u;
        px = 2 * x + int(sampleIdx & 1u);
        py = 2 * y + int(sampleIdx >> 1u);
    }
    const v2 uv{(float(px) + 0.5f) * cam.ivw, (float(py) + 0.5f) * cam.ivh};
    const v3 normalVS  = mul_dir(xyz(ld<v4>(normalTex, px, py)), cam.view);
    const float rough  = ld<rough_t>(roughnessTex, px, py);
    const bool mirror  = rough < 0.01f; // IsMirrorReflection
    const int  mdm     = mirror ? 0 : int(k.MostDetailedMip);
    const v2   mipRes0 = screen * fdiv(1.0f, float(1 << mdm));
    const v3   originSS{uv.x, uv.y, load_hiz<false>(hiz, int(uv.x * mipRes0.x), int(uv.y * mipRes0.y), mdm)};
    const v3   originVS = screen_xy_depth_to_view_space(originSS, cam.proj);

    // SampleReflectionVector :254-278 (GGX VNDF, spherical caps)
    const v3 view = -normalize(originVS);
    v3 dirVS;
    float pdf;
    {
        const float alpha = rough * rough;
        const v3 N = normalVS;
        const v3 T = normalize(cross(N, fabsf(N.y) > 0.5f ? v3{1.0f, 0.0f, 0.0f} : v3{0.0f, 1.0f, 0.0f}));
        const v3 B = cross(T, N);
        v2 xi = ld<v2>(noiseXY, x & 127, y & 127);
        xi.y  = lerpf(xi.y, 0.0f, k.GGXImportanceSampleBias);
        const v3 viewTS{dot(T, view), dot(B, view), dot(N, view)};
        const v3 micro  = smith_ggx_sample_visible_normal_sc(viewTS, alpha, alpha, xi.x, xi.y);
        const v3 sampTS = reflect(-viewTS, micro);
        const float NdotV = viewTS.z, NdotH = micro.z;
        const float D  = normal_distribution_ggx(NdotH, alpha);
        const float G1 = smith_ggx_masking(NdotV, alpha);
        pdf   = fdiv(G1 * D, 4.0f * NdotV + SSR_FLT_EPS);
        dirVS = sampTS.x * T + sampTS.y * B + sampTS.z * N;
    }
    const v3 dirSS = project_position(originVS + dirVS, cam.proj) - originSS; // ProjectDirection
    tail = R4Tail{uv, mul_dir(dirVS, cam.viewInv), originVS, pdf};

    // the head of HierarchicalRaymarch (:139-150) and InitialAdvanceRay (:66-86)
    m.origin = originSS;
    m.dir    = dirSS;
    m.invDir = v3{dirSS.x != 0.0f ? fdiv(1.0f, dirSS.x) : SSR_FLT_MAX, dirSS.y != 0.0f ? fdiv(1.0f, dirSS.y) : SSR_FLT_MAX, dirSS.z != 0.0f ? fdiv(1.0f, dirSS.z) : SSR_FLT_MAX};
    m.loMin  = (mdm + 1) * kEntry;
    m.lo     = m.loMin;
    m.away   = REV ? dirSS.z < 0.0f : dirSS.z > 0.0f;
    v2 uvOffset = (0.005f * float(1 << mdm)) / screen;
    uvOffset.x = dirSS.x < 0.0f ? -uvOffset.x : uvOffset.x;
    uvOffset.y = dirSS.y < 0.0f ? -uvOffset.y : uvOffset.y;
    m.uvOffset    = uvOffset;
    m.floorOffset = v2{dirSS.x < 0.0f ? 0.0f : 1.0f, dirSS.y < 0.0f ? 0.0f : 1.0f};
    {
        const HizLevel L = hiz.lv[mdm + 1];
        const v2 mipRes{L.res.x, L.res.y}, invMipRes{L.res.z, L.res.w};
        const v2 mp = mipRes * mk2(originSS.x, originSS.y);
        v2 plane{floorf(mp.x) + m.floorOffset.x, floorf(mp.y) + m.floorOffset.y};
        plane = plane * invMipRes + uvOffset;
        const v2 t{plane.x * m.invDir.x - originSS.x * m.invDir.x, plane.y * m.invDir.y - originSS.y * m.invDir.y};
        m.curT = fminf(t.x, t.y);
        m.pos  = originSS + m.curT * dirSS;
    }
    return true;
}

// Everything of the one-ray kernel behind the march.
template <bool PREV, bool REV>
MIFX_D void r4_finish(const HizLds& hiz, const Img& radiance, const Img& normalTex, const Img& motionTex, const Img& outSpec, const Img& outDirPdf, const Img& hitCoords, const CamK& cam,
                      const SsrK& k, int x, int y, v3 hitSS, const R4Tail& tail, int localBegin, int localEnd)
{
    const v2 screen{cam.vw, cam.vh};
    const v3 hitVS = screen_xy_depth_to_view_space(hitSS, cam.proj);
    v2 hitPrev{hitSS.x, hitSS.y};
    if (PREV)
    {
        const v2 mv = ld_zero_v2(motionTex, int(screen.x * hitSS.x), int(screen.y * hitSS.y)); // LoadMotion :56-59
        hitPrev = v2{hitSS.x - mv.x * 0.5f, hitSS.y - mv.y * -0.5f};
    }
    const float confidence = validate_hit<PREV, REV, false>(hiz, normalTex, hitSS, hitPrev, tail.uv, tail.dirWS, screen, k.DepthBufferThickness, cam.proj);
    v3 refl = mk3(0.0f);
    unsigned where = 0xffffffffu;
    if (confidence > 0.0f)
    {
        const int  rx = int(screen.x * hitPrev.x), ry = int(screen.y * hitPrev.y);
        const bool in = rx >= 0 && ry >= 0 && rx < radiance.w && ry < radiance.h;
        if (hitCoords.p != nullptr && (ry < localBegin || ry >= localEnd)) where = in ? unsigned(rx) | (unsigned(ry) << 16) : 0xffffffffu;
        else if (in) refl = xyz(ld<v4>(radiance, rx, ry));
    }
    if (hitCoords.p != nullptr) st<float>(hitCoords, x, y, __uint_as_float(where));
    st<v4>(outSpec, x, y, mk4(refl, confidence));
    st<v4>(outDirPdf, x, y, mk4(tail.dirWS * length(hitVS - tail.originVS), tail.pdf));
}

// One step of AdvanceRay (:88-137) + the level update (:171-179) for a ray that is still marching; `act` false leaves the state as it is.
template <bool REV>
MIFX_D void r4_step(R4March& m, const HizLevel& L, v2 mp, float surfaceDepth, bool act)
{
    constexpr int kEntry = int(sizeof(HizLevel));
    const v2 invMipRes{L.res.z, L.res.w};
    v2 plane{floorf(mp.x) + m.floorOffset.x, floorf(mp.y) + m.floorOffset.y};
    plane = plane * invMipRes + m.uvOffset;
    v3 t{plane.x * m.invDir.x - m.origin.x * m.invDir.x, plane.y * m.invDir.y - m.origin.y * m.invDir.y, surfaceDepth * m.invDir.z - m.origin.z * m.invDir.z};
    t.z = m.away ? t.z : SSR_FLT_MAX;
    const float tmin = fminf(fminf(t.x, t.y), t.z);
    const bool  above = REV ? surfaceDepth < m.pos.z : surfaceDepth > m.pos.z;
    const bool  skipped = __float_as_uint(tmin) != __float_as_uint(t.z) && above;
    m.curT = (above && act) ? tmin : m.curT;
    m.pos  = m.origin + m.curT * m.dir;
    const int next = min(m.lo + (skipped ? kEntry : -kEntry), (SSR_MAX_MIP + 1) * kEntry);
    m.lo = act ? next : m.lo;
}

#ifndef MIFX_R4X2_WAVES
#define MIFX_R4X2_WAVES 6
#endif
template <bool PREV, bool REV>
__global__ __launch_bounds__(256) MIFX_WAVES_OPT(MIFX_R4X2_WAVES) MIFX_R4_SGPR_CAP void ssr_intersection2_kernel(Img radiance, Img normalTex, Img roughnessTex, Img noiseXY, HizSlab hizSlab, Img mask, Img motionTex,
                                                                                                       Img outSpec, Img outDirPdf, CamK cam, SsrK k, Img hitCoords, int localBegin, int localEnd)
{
    __shared__ HizLevel hizLv[SSR_MAX_MIP + 2];
    __shared__ float    parked[2 * kR4TailFloats * 256];
    if (threadIdx.x < unsigned(SSR_MAX_MIP + 2))
    {
        const unsigned lv = threadIdx.x == 0u ? 0u : threadIdx.x - 1u; // entry 0 = a second copy of level 0
        const float s = fdiv(1.0f, float(1 << int(lv)));
        const v2    r{cam.vw * s, cam.vh * s};
        hizLv[threadIdx.x] = HizLevel{uint4{hizSlab.offset[lv], hizSlab.pitch[lv], hizSlab.w[lv], hizSlab.h[lv]}, v4{r.x, r.y, fdiv(1.0f, r.x), fdiv(1.0f, r.y)}};
    }
    __syncthreads();
    const HizLds hiz{__builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hizSlab.base), 0, int(hizSlab.bytes), 0x00020000), hizLv,
                     __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(hizSlab.base), 0, int(hizSlab.bytes), 0x00020000)};
    // wave w of the workgroup: the 16 x 8 block at x = 64 blockIdx.x + 16 w; lane l: pixel (l & 7, l >> 3) of its left tile (ray 0) and of its right tile (ray 1)
    const int t = int(threadIdx.x), lane = t & 63;
    const int x0 = int(blockIdx.x) * 64 + (t >> 6) * 16 + (lane & 7), x1 = x0 + 8;
    const int y  = int(blockIdx.y) * 8 + (lane >> 3) + outSpec.y0;
    float* const slot0 = parked + t;
    float* const slot1 = parked + kR4TailFloats * 256 + t;
    R4March a, b;
    {
        R4Tail tail;
        (void)r4_setup<REV>(hiz, normalTex, roughnessTex, noiseXY, mask, outSpec, outDirPdf, hitCoords, cam, k, x0, y, a, tail);
        r4_park(slot0, tail);
        (void)r4_setup<REV>(hiz, normalTex, roughnessTex, noiseXY, mask, outSpec, outDirPdf, hitCoords, cam, k, x1, y, b, tail);
        r4_park(slot1, tail);
    }
    const bool hasA = a.lo >= a.loMin, hasB = b.lo >= b.loMin;
    // the march of both rays in lock step: both taps are issued, then both steps taken.  A ray that has finished (or a slot without a ray) taps texel 0 of the slab --
    // all such lanes share one cache line -- and keeps its state.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    auto entry = [&](int o) {
        const char* p = reinterpret_cast<const char*>(hizLv) + o;
        const u32x4   ad = *reinterpret_cast<const u32x4*>(p);
        const mifx_f4 rs = *reinterpret_cast<const mifx_f4*>(p + 16);
        return HizLevel{uint4{ad.x, ad.y, ad.z, ad.w}, v4{rs.x, rs.y, rs.z, rs.w}};
    };
    auto tap_offset = [&](const HizLevel& L, v2 mp, bool act, bool& inside) {
        const int tx = int(mp.x), ty = int(mp.y);
        inside = unsigned(tx) < L.addr.z && unsigned(ty) < L.addr.w;
        unsigned row, off;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row) : "v"(ty), "v"(L.addr.y), "v"(L.addr.x));
        asm("v_lshl_add_u32 %0, %1, 2, %2" : "=v"(off) : "v"(tx), "v"(row));
        return act ? off : 0u;
    };
    HizLevel La = entry(a.lo), Lb = entry(b.lo);
    unsigned idx = 0u;
    bool actA = hasA && idx < k.MaxTraversalIntersections, actB = hasB && idx < k.MaxTraversalIntersections;
    while (actA || actB)
    {
        const v2 mpA = mk2(La.res.x, La.res.y) * mk2(a.pos.x, a.pos.y), mpB = mk2(Lb.res.x, Lb.res.y) * mk2(b.pos.x, b.pos.y);
        bool inA, inB;
        const unsigned offA = tap_offset(La, mpA, actA, inA), offB = tap_offset(Lb, mpB, actB, inB);
        const float rawA = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hiz.rsrc, int(offA), 0, 0));
        const float rawB = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(hiz.rsrc, int(offB), 0, 0));
        r4_step<REV>(a, La, mpA, inA ? rawA : 0.0f, actA);
        La = entry(a.lo);
        r4_step<REV>(b, Lb, mpB, inB ? rawB : 0.0f, actB);
        Lb = entry(b.lo);
        ++idx;
        actA = a.lo >= a.loMin && idx < k.MaxTraversalIntersections;
        actB = b.lo >= b.loMin && idx < k.MaxTraversalIntersections;
    }
    if (hasA) r4_finish<PREV, REV>(hiz, radiance, normalTex, motionTex, outSpec, outDirPdf, hitCoords, cam, k, x0, y, a.pos, r4_unpark(slot0), localBegin, localEnd);
    if (hasB) r4_finish<PREV, REV>(hiz, radiance, normalTex, motionTex, outSpec, outDirPdf, hitCoords, cam, k, x1, y, b.pos, r4_unpark(slot1), localBegin, localEnd);
}
#endif // MIFX_R4_TWO_RAYS

static const dim3 kBlock(64, 4, 1);
#define MIFX_LAUNCH_END()              \
    MIFX_HIP_CHECK(hipGetLastError()); \
    return MIFX_OK

mifx_status launch_ssr_intersection(hipStream_t s, Img radiance, Img normal, Img roughness, Img noiseXY, const HizSlab& hiz, Img mask, Img motion, Img outSpec, Img outDirPdf,
                                    const CamK& cam, const mifx_ssr_attribs& a, bool previousFrame, bool halfResolution, Img hitCoords, int localBegin, int localEnd)
{
    const bool rev = cam.reversedDepth != 0;
    const SsrK k   = make_k(a, rev, halfResolution);
#ifndef MIFX_R4_BLOCK
#define MIFX_R4_BLOCK 256 // (measured late in round 2: one or two 8x8 tiles per workgroup, -DMIFX_R4_BLOCK=64 / 128, are 2-4 % slower)
#endif
    const dim3 r4grid(xcd_grid_x((outSpec.w + MIFX_R4_BLOCK / 8 - 1) / (MIFX_R4_BLOCK / 8)), (window_rows(outSpec) + 7) / 8, 1);
    // Experiment knob (MIFX_R4_LDS_PAD=<bytes>): unused dynamic LDS per workgroup, which bounds the workgroups a CU holds (160 KB / pad) and so leaves wave slots to a
    // kernel that runs beside the march on another stream.
    static const unsigned ldsPad = occupancy_pad_from_env("MIFX_R4_LDS_PAD");
#if MIFX_R4_TWO_RAYS
    // the experiment build: MIFX_R4_RAYS=2 in the environment selects the two-rays-per-lane kernel (both write the same bits)
    static const bool twoRays = []() { const char* e = std::getenv("MIFX_R4_RAYS"); return e != nullptr && std::atoi(e) == 2; }();
    const dim3 r4grid2((outSpec.w + 63) / 64, (window_rows(outSpec) + 7) / 8, 1);
#define MIFX_R4_LAUNCH(P, R)                                                                                                                                                               \
    do {                                                                                                                                                                                   \
        if (twoRays) hipLaunchKernelGGL((ssr_intersection2_kernel<P, R>), r4grid2, dim3(256, 1, 1), ldsPad, s, radiance, normal, roughness, noiseXY, hiz, mask, motion, outSpec, outDirPdf, cam, k, hitCoords, localBegin, localEnd); \
        else hipLaunchKernelGGL((ssr_intersection_kernel<P, R>), r4grid, dim3(MIFX_R4_BLOCK, 1, 1), ldsPad, s, radiance, normal, roughness, noiseXY, hiz, mask, motion, outSpec, outDirPdf, cam, k, hitCoords, localBegin, localEnd); \
    } while (0)
#else
#define MIFX_R4_LAUNCH(P, R)                                                                                                                                                       \
    do {                                                                                                                                                                           \
        if (hiz.base0 != nullptr) hipLaunchKernelGGL((ssr_intersection_kernel<P, R, true>), r4grid, dim3(MIFX_R4_BLOCK, 1, 1), ldsPad, s, radiance, normal, roughness, noiseXY, hiz, mask, motion, outSpec, outDirPdf, cam, k, hitCoords, localBegin, localEnd); \
        else hipLaunchKernelGGL((ssr_intersection_kernel<P, R>), r4grid, dim3(MIFX_R4_BLOCK, 1, 1), ldsPad, s, radiance, normal, roughness, noiseXY, hiz, mask, motion, outSpec, outDirPdf, cam, k, hitCoords, localBegin, localEnd); \
    } while (0)
#endif
    if (previousFrame) { if (rev) MIFX_R4_LAUNCH(true, true); else MIFX_R4_LAUNCH(true, false); }
    else { if (rev) MIFX_R4_LAUNCH(false, true); else MIFX_R4_LAUNCH(false, false); }
#undef MIFX_R4_LAUNCH
    MIFX_LAUNCH_END();
}
} // namespace mifx
