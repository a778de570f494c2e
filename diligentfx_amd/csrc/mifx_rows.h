// mifx_rows.h -- row ranges for row-band sharding.
//
// A rank owns the rows [band_b, band_e) of the final image.  Every pass is launched on the rows its consumers need (its row window), which
// is the band grown by the reach of everything downstream; rows outside a window keep stale data and are never read.  Reaches below are
// upper bounds of the tap footprints in the kernels (cited); windows are clipped to the frame, so the whole-frame case reproduces the
// unsharded launch exactly.
#pragma once
#include "mifx_host.h"

namespace mifx
{
struct Rows
{
    int b, e; // [b, e)
    bool empty() const { return e <= b; }
};
inline Rows rows_clip(Rows r, int h) { return Rows{r.b < 0 ? 0 : r.b, r.e > h ? h : r.e}; }
inline Rows rows_expand(Rows r, int g, int h) { return rows_clip(Rows{r.b - g, r.e + g}, h); }
inline Rows rows_hull(Rows a, Rows b) { return Rows{a.b < b.b ? a.b : b.b, a.e > b.e ? a.e : b.e}; }
inline Rows rows_align(Rows r, int a, int h) { return rows_clip(Rows{(r.b / a) * a, ((r.e + a - 1) / a) * a}, h); }
// rows of the next coarser level (half resolution, hl rows) that cover the rows r of the finer level, grown by g coarse rows
inline Rows rows_coarser(Rows r, int g, int hl) { return rows_clip(Rows{r.b / 2 - g, (r.e + 1) / 2 + g}, hl); }
// rows of the next finer level (hf rows) that the rows r of the coarser level are computed from, grown by g fine rows
inline Rows rows_finer(Rows r, int g, int hf) { return rows_clip(Rows{2 * r.b - g, 2 * r.e + g}, hf); }
inline Img  win(Img im, Rows r) { return rows_of(im, r.b, r.e); }
inline bool rows_contain(Rows outer, Rows inner) { return inner.empty() || (outer.b <= inner.b && inner.e <= outer.e); }
} // namespace mifx
