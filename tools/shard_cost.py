#!/usr/bin/env python3
"""Compute-side cost of row-band sharding, measured on ONE GPU: the time one rank of an N-rank job spends on its band (ghost rows included,
exchanges left out: the planes then hold stale rows, which does not change the work) against 1/N of the whole-frame time.

    python tools/shard_cost.py [--width 7680 --height 4320 --world 8 --steps 10]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import tiling  # noqa: E402


def timed(fn, steps, first):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        fn(first + i)
    timed.issue_ms = (time.perf_counter() - t0) / steps * 1e3  # host time to enqueue a frame (no waiting)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=7680)
    p.add_argument("--height", type=int, default=4320)
    p.add_argument("--world", type=int, default=8)
    p.add_argument("--steps", type=int, default=10)
    p.add_argument("--ranks", type=int, nargs="*", default=None, help="ranks to time (default: first, middle, last)")
    p.add_argument("--whole-ms", type=float, default=0.0, help="skip the whole-frame timing and use this value")
    p.add_argument("--orbit-frames", type=int, default=6, help="resident camera positions (2.3 GB each at 7680x4320)")
    p.add_argument("--weighted", action="store_true", help="cost-weighted band heights (tiling.cost_weighted_cuts) instead of equal bands")
    p.add_argument("--sky-cost", type=float, default=None, help="relative cost of a row of background for --weighted (default: the library's)")
    p.add_argument("--cuts", type=int, nargs="*", default=None, help="explicit row boundaries (world + 1 values) instead of equal or weighted bands")
    p.add_argument("--classes", action="store_true", help="print, per band, the rows of sky / geometry / reflection samples within the band + 60 ghost rows (for refitting tiling's cost model)")
    p.add_argument("--reflective-cost", type=float, default=-1.0, help="relative cost of a reflection sample for --weighted (default: the library's; 0 = two-class model)")
    p.add_argument("--overlap", type=int, default=3, help="mifx_chain_set_overlap of the band's chain: 3 = the sharded frame's three lanes (what bench.py --gpus N runs), 2 = two lanes "
                   "(phase 3 beside the next frame), 0 = the phases back to back on one stream (rounds 1-4)")
    p.add_argument("--refine", type=int, default=0, help="rounds of tiling.refine_cuts: every band is timed, the cuts move towards equal measured times, all bands are timed again "
                   "(what bench.py --gpus N does before its warm-up: TiledChain.calibrate_cuts)")
    a = p.parse_args()
    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    r = tiling.TiledChain(0, tables["sobol_256d"], tables["scrambling_tile"], 0, 1, a.width, a.height)
    r.build_inputs(n_frames=a.orbit_frames)
    # The whole frame on this one GPU, in the same stream mode as the band (mifx_chain_set_overlap: lanes across frames for both, or one stream for both) -- the figure the
    # speed-up is quoted against -- and on one stream (what rounds 1 - 4 quoted against, when the band ran on one stream as well).
    whole_one_stream = None
    if a.whole_ms > 0.0:
        whole = a.whole_ms
    else:
        warm = 2 * a.orbit_frames
        for i in range(warm):
            r.step(i)
        whole = whole_one_stream = timed(r.step, a.steps, warm)
        if a.overlap > 0:
            r.chain.set_overlap(a.overlap)
            for i in range(warm):
                r.step(warm + a.steps + i)
            whole = timed(r.step, a.steps, 2 * warm + a.steps)
            # ... and with two frames in flight (mifx_chain_set_overlap 5: the unsharded chain's best mode since the second session of round 6; the band has no such mode)
            r.chain.set_overlap(5)
            for i in range(warm):
                r.step(2 * warm + 2 * a.steps + i)
            a.whole_mode4 = timed(r.step, a.steps, 3 * warm + 2 * a.steps)
            r.chain.set_overlap(0)
    max_motion = int(max(max(float(f["motion_fwd"][..., 1].abs().max()), float(f["motion_bwd"][..., 1].abs().max())) for f in r.frames) * 0.5 * a.height) + 2
    print(f"{a.width}x{a.height}: whole frame {whole:.3f} ms" + (f" (mifx_chain_set_overlap {a.overlap}; one stream: {whole_one_stream:.3f} ms; two frames in flight, mode 5: {getattr(a, 'whole_mode4', 0.0):.3f} ms)" if whole_one_stream and a.overlap > 0 else "") +
          f"; max motion {max_motion} rows")
    a.whole_one_stream = whole_one_stream
    rows = a.height // a.world
    rc = "default" if a.reflective_cost < 0 else (None if a.reflective_cost == 0 else a.reflective_cost)
    cuts = tiling.band_cuts(r.frames[0], r.chain.ssr_attribs, a.world, min(192, rows), sky_cost=a.sky_cost, reflective_cost=rc) if a.weighted else tuple(i * rows for i in range(a.world + 1))
    if a.cuts:
        assert len(a.cuts) == a.world + 1 and a.cuts[0] == 0 and a.cuts[-1] == a.height, a.cuts
        cuts = tuple(a.cuts)
    print("  cuts", list(cuts))
    r.chain.set_overlap(a.overlap)
    for rnd in range(a.refine + 1):
        if rnd > 0:
            cuts = tiling.refine_cuts(cuts, times, a.height, min(192, a.height // a.world))
            print(f"  -- refined from the measured band times (round {rnd}): cuts {list(cuts)}")
        times = measure(a, r, cuts, max_motion, whole, all_ranks=a.refine > 0)
    r.chain.set_row_band(0, 0, 0)
    r.chain.set_overlap(0)


def measure(a, r, cuts, max_motion, whole, all_ranks):
    worst, times = 0.0, []
    if a.classes:
        f0 = r.frames[0]
        is_geom = f0["depth"] < 1.0 - 1e-6
        geom_row = is_geom.float().mean(dim=1).double().cpu().numpy()
        refl_row = (is_geom & (f0["material"][..., int(r.chain.ssr_attribs.RoughnessChannel)].float() <= float(r.chain.ssr_attribs.RoughnessThreshold))).float().mean(dim=1).double().cpu().numpy()
    for rank in (range(a.world) if all_ranks else (a.ranks if a.ranks else sorted({0, a.world // 2, a.world - 1}))):
        rows = cuts[rank + 1] - cuts[rank]
        r.chain.set_row_band(cuts[rank], cuts[rank + 1], max_motion)

        bound = {}

        def band_step(i, bound=bound):
            k, kp = r.orbit_position(i)  # the same forwards-and-back walk over the resident orbit as the whole-frame run
            b = bound.get((k, kp))
            if b is None:
                b = bound[(k, kp)] = r.chain.bind_frame(2000 + i, r._frame_view(k, kp), r.ibl, r.shade, r.out)
            b[0].frame.Index = 2000 + i
            r.chain.execute_band(b)

        warm = 2 * a.orbit_frames  # one full walk of the orbit: every frame descriptor of the band is bound (and every kernel variant loaded) before the clock starts
        for i in range(warm):
            band_step(i)
        t = timed(band_step, a.steps, warm)
        info = r.chain.shard_info(r.chain.bind_frame(1, r._frame_view(1, 0), r.ibl, r.shade, r.out))
        worst = max(worst, t)
        times.append(t)
        if a.classes:
            lo, hi = max(cuts[rank] - 60, 0), min(cuts[rank + 1] + 60, a.height)
            g, rf = float(geom_row[lo:hi].sum()), float(refl_row[lo:hi].sum())
            print(f"  CLASSES rank {rank} rows {hi - lo} sky {hi - lo - g:.1f} geometry {g - rf:.1f} reflective {rf:.1f} ms {t:.4f}")
        print(f"  rank {rank}/{a.world}: band of {rows} rows {t:.3f} ms = {t / (whole / a.world):.2f}x of whole/N  -> compute-side efficiency {whole / a.world / t:.2f}"
              f"  (halos without Bloom's level-0 exchange: taa {info.halo_taa} ssr {info.halo_ssr} ssao {info.halo_ssao} rows)")
    print(f"  slowest band {worst:.3f} ms -> compute-side speed-up {whole / worst:.2f}x on {a.world} GPUs" +
          (f"  ({a.whole_one_stream / worst:.2f}x against the whole frame on one stream" + (f"; {a.whole_mode4 / worst:.2f}x against it with two frames in flight" if getattr(a, "whole_mode4", None) else "") + ")"
           if getattr(a, "whole_one_stream", None) and a.overlap > 0 else ""))
    return times


if __name__ == "__main__":
    main()
