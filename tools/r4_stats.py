#!/usr/bin/env python3
"""How many march steps do the rays of SSR's R4 take, and how much of a wave's march is spent waiting for its longest ray?  Needs the statistics build of the library
(the step count of every ray stored in place of the pdf):

    python -c "from diligentfx_amd import build as B; B.build_variant('/tmp/libmifx_r4stats.so', '/tmp/obj_r4stats', ['-DMIFX_R4_STATS=1'])"   (here)
    MIFX_LIB_PATH=diligentfx_amd/variants/r4stats.so python tools/r4_stats.py                                                         (GPU box)

A wave of R4 holds one 8x8 tile (tiled_xy, mifx_device.h); it runs the march loop max(steps) times, a lane is useful for its own steps."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

from diligentfx_amd import tiling  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--frames", type=int, default=30)
    a = p.parse_args()
    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    r = tiling.TiledChain(0, tables["sobol_256d"], tables["scrambling_tile"], 0, 1, a.width, a.height)
    r.build_inputs(n_frames=24)
    for _ in range(a.frames):
        r.step()
    ssr = r.chain.effect("ssr")
    packed = ssr.get_intermediate("ray_dir_pdf")[..., 3].float().to(torch.int64)  # steps + 256 x (steps at levels >= 5) + 65536 x (steps at levels >= 4)
    steps, coarse5, coarse4 = (packed & 255).float(), ((packed >> 8) & 255).float(), (packed >> 16).float()
    mask = ssr.get_intermediate("mask") != 0
    H, W = steps.shape
    h8, w8 = H // 8 * 8, W // 8 * 8
    t = steps[:h8, :w8].reshape(h8 // 8, 8, w8 // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
    m = mask[:h8, :w8].reshape(h8 // 8, 8, w8 // 8, 8).permute(0, 2, 1, 3).reshape(-1, 64)
    active = m.any(dim=1)
    tmax = t.max(dim=1).values
    useful, issued = float(t.sum()), float(tmax[active].sum()) * 64.0
    print(f"{W}x{H} after {a.frames} frames of the bench orbit")
    print(f"rays: {float(mask.float().mean()):.4f} of the texels; waves with at least one ray: {float(active.float().mean()):.4f} of the tiles; "
          f"full waves (64 rays): {float(m.all(dim=1).float().mean()):.4f}")
    rays = steps[mask]
    q = torch.quantile(rays, torch.tensor([0.1, 0.25, 0.5, 0.75, 0.9, 0.99], device=rays.device))
    print(f"steps per ray: mean {float(rays.mean()):.1f}, percentiles 10/25/50/75/90/99: {' '.join(f'{float(v):.0f}' for v in q)}, max {float(rays.max()):.0f}")
    print(f"steps per wave (its longest ray): mean {float(tmax[active].mean()):.1f}")
    print(f"steps at hierarchy levels >= 5: {float(coarse5[mask].sum()) / float(rays.sum()):.3f} of all steps; at levels >= 4: {float(coarse4[mask].sum()) / float(rays.sum()):.3f}")
    print(f"lane utilisation of the march: {useful / issued:.3f} (useful lane-steps / 64 x the wave's steps)")
    # what a compaction inside the 256-thread workgroup (4 tiles) would see: the rays of a workgroup sorted into waves
    for g, name in ((4, "workgroup of 4 tiles, rays compacted"),):
        n = t.shape[0] // g * g
        tg, mg = t[:n].reshape(-1, 64 * g), m[:n].reshape(-1, 64 * g)
        cnt = mg.sum(dim=1)
        waves = (cnt + 63) // 64
        print(f"{name}: {float(waves.sum()) / float(active.sum()):.3f} of today's marching waves")


if __name__ == "__main__":
    main()
