"""tiling.refine_cuts: band heights fed back from measured band times (host logic of bench.py --gpus N; no GPU)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_refine_cuts_properties():
    from diligentfx_amd import tiling

    h, world, min_rows = 4320, 8, 192
    cuts = (0, 1323, 1875, 2331, 2723, 3133, 3525, 3894, 4320)
    # equal times: nothing moves (up to rounding)
    same = tiling.refine_cuts(cuts, [1.3] * world, h, min_rows)
    assert all(abs(a - b) <= 1 for a, b in zip(same, cuts))
    # a slow band shrinks, its neighbours grow; the cuts stay ordered, cover the frame and respect the minimum height
    times = [1.28, 1.30, 1.31, 1.35, 1.34, 1.28, 1.28, 1.33]
    new = tiling.refine_cuts(cuts, times, h, min_rows)
    assert new[0] == 0 and new[-1] == h and all(b - a >= min_rows for a, b in zip(new, new[1:]))
    assert (new[4] - new[3]) < (cuts[4] - cuts[3]) and (new[1] - new[0]) > (cuts[1] - cuts[0])
    # a pure function of its arguments: every rank arrives at the same cuts
    assert new == tiling.refine_cuts(cuts, list(times), h, min_rows)
    # the model's fixed point: times proportional to rows above the fixed part -> equal bands
    eq = tiling.refine_cuts((0, 100, 400, 1000), [0.17 + 0.1, 0.17 + 0.3, 0.17 + 0.6], 1000, 10)
    assert abs(eq[1] - 333) <= 1 and abs(eq[2] - 667) <= 1
    # damping moves part of the way
    half = tiling.refine_cuts(cuts, times, h, min_rows, damping=0.5)
    assert all(min(a, b) <= c <= max(a, b) for a, b, c in zip(cuts, new, half))
