// mifx_pyramid.h -- several levels of a 2x2-reduction pyramid from one kernel.
//
// R1 (closest depth), A2 (prefiltered depth) and A6 (box-filtered AO / depth) build their levels one 2x2 reduction at a time; as separate
// dispatches the small levels are pure launch latency (14 dispatches, ~140 us of a 2.5 ms frame).  While the source dimensions stay even, a
// texel of level k + j depends only on its own 2^j x 2^j block of level k, so one workgroup can take a 32x32 block of level k down to the
// 2x2 block of level k + 4 through LDS, writing every level on the way.  Each texel is produced by the same reduction of the same four
// stored values as in the level-by-level kernels (tap order (0,0), (0,1), (1,0), (1,1) as in the reference loops), so results are
// bit-identical; levels whose source has an odd dimension (3-wide taps) keep the single-level kernels.
#pragma once
#include "mifx_host.h"

namespace mifx
{
// Two horizontally adjacent single-channel texels (x even) as one 8-byte access: a 2x2 source block is two such loads, and the 16 lanes of a row of the
// workgroup then read 128 contiguous bytes per instruction instead of every other dword of them twice.  Needs an 8-byte aligned base and pitch
// (pair_aligned(): uniform per launch; planes of the library always are, a caller's depth buffer may not be).
MIFX_D v2   ld_pair(const Img& im, int x, int y) { return GlobalAccess<v2>::load(im.p + size_t(y) * im.pitch + size_t(x) * 4u); }
MIFX_D void st_pair(const Img& im, int x, int y, v2 v) { GlobalAccess<v2>::store(im.p + size_t(y) * im.pitch + size_t(x) * 4u, v); }
MIFX_HD bool pair_aligned(const Img& im) { return (reinterpret_cast<uintptr_t>(im.p) & 7u) == 0u && (im.pitch & 7) == 0; }

// OP: T (value type), void quad(x, y, a, b, c, d): the source texels (2x, 2y), (2x, 2y + 1), (2x + 1, 2y), (2x + 1, 2y + 1), T reduce(T, T, T, T), bool inside(level, x, y), void store(level, x, y, T),
// T stored(T) (what a store + load of a produced texel returns: the identity unless the level is kept in a narrow format), int first_row()
// with level = 1 .. nl relative to the source.  Launch: block (256, 1, 1), grid (ceil(w1 / 16), ceil(rows1 / 16)), w1 = width and rows1 =
// rows of the window of level 1, whose first row -- first_row() -- must be a multiple of 2^(nl - 1) (8: one row of the last level): a workgroup then still holds the
// whole 2^nl x 2^nl source block of every texel it produces.  (Until round 5 the window had to start on a multiple of 16 rows of level 1.)
template <class OP> MIFX_D void pyramid_reduce_levels(const OP& op, int nl)
{
    using T = typename OP::T;
    __shared__ T lds[16 * 16 + 8 * 8 + 4 * 4 + 2 * 2];
    const int tid = int(threadIdx.x);
    // row window: the launch covers the level-1 rows [r1, ...) and the rows of the deeper levels below them (op.first_row(): 0 for whole images); op.inside() applies
    // the end of each level's window
    const int r1 = op.first_row(), yb = int(blockIdx.y);
    {
        const int lx = tid & 15, ly = tid >> 4, x = int(blockIdx.x) * 16 + lx, y = r1 + yb * 16 + ly;
        T v{};
        if (op.inside(1, x, y))
        {
            T a, b, c, d;
            op.quad(x, y, a, b, c, d);
            v = op.stored(op.reduce(a, b, c, d)); // (stored(): the value a consumer reads back -- the next level is reduced from the stored level, as in the per-level passes)
            op.store(1, x, y, v);
        }
        lds[ly * 16 + lx] = v;
    }
    T*  src     = lds;
    int srcSide = 16;
    for (int l = 2; l <= nl; ++l)
    {
        __syncthreads();
        const int side = srcSide >> 1;
        T*        dst  = src + srcSide * srcSide;
        if (tid < side * side)
        {
            const int lx = tid % side, ly = tid / side, x = int(blockIdx.x) * side + lx, y = (r1 >> (l - 1)) + yb * side + ly;
            const T*  p  = src + (2 * ly) * srcSide + 2 * lx;
            const T   v  = op.stored(op.reduce(p[0], p[srcSide], p[1], p[srcSide + 1]));
            if (op.inside(l, x, y)) op.store(l, x, y, v);
            dst[ly * side + lx] = v;
        }
        src     = dst;
        srcSide = side;
    }
}
// (pyramid_fusable_levels: mifx_host.h -- the host objects ask as well)
} // namespace mifx
