#!/usr/bin/env python3
"""mifx_pbr_shade_execute_layers at 3840x2160: the default shade, each material layer alone, all five, and all five with two shadow-mapped lights.  HIP-event time per call
(cube aprons included), algorithmic bytes per pixel = the planes the permutation reads and writes.

    python tools/layers_timing.py [--width 3840 --height 2160 --steps 20]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
from diligentfx_amd import api, binding as B, synth  # noqa: E402


def timed(fn, steps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / steps * 1e3


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--ab", action="store_true", help="run twice in child processes: the instances compiled per layer set (default) and MIFX_LAYERS_GENERIC=1 (the run-time-set instance for every set)")
    a = p.parse_args()
    if a.ab:
        import subprocess

        for v in ("0", "1"):
            print(f"==== MIFX_LAYERS_GENERIC={v}", flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "--width", str(a.width), "--height", str(a.height), "--steps", str(a.steps)],
                           env=dict(os.environ, MIFX_LAYERS_GENERIC=v), check=True)
        return
    w, h = a.width, a.height
    ctx = api.PostFXContext(0)
    ctx.set_static_ibl(True)
    dev = ctx.device
    env = synth.make_sky_cube(64, dev)
    ibl = api.precompute_ibl(ctx, env, lut_size=128, irradiance_size=16, prefiltered_size=128, lut_samples=128, diffuse_samples=256, specular_samples=64)
    sa = synth.make_lights()
    sa.PrefilteredCubeLastMip = float(len(ibl.pre) - 1)
    f = synth.make_frame(synth.Scene(), 3, w, h, dev)
    g = {k: f[k] for k in ("base_color", "normal", "material", "depth")}
    gen = torch.Generator(device=dev).manual_seed(5)
    r = lambda *s: torch.rand(*s, device=dev, generator=gen)  # noqa: E731
    n = f["normal"][..., :3]
    t = r(h, w, 3) - 0.5
    t = t - n * (t * n).sum(-1, keepdim=True)
    z = torch.zeros(h, w, 1, device=dev)
    ang = 6.2831853 * r(h, w, 1)
    planes = {"clearcoat": torch.cat([r(h, w, 1), 0.05 + 0.95 * r(h, w, 1), z, z], -1), "clearcoat_normal": torch.cat([torch.nn.functional.normalize(n + 0.3 * (r(h, w, 3) - 0.5), dim=-1), z], -1),
              "sheen": torch.cat([r(h, w, 3), 0.05 + 0.95 * r(h, w, 1)], -1), "anisotropy": torch.cat([torch.cos(ang), torch.sin(ang), r(h, w, 1), z], -1),
              "tangent": torch.cat([torch.nn.functional.normalize(t, dim=-1), z], -1), "iridescence": torch.cat([r(h, w, 1), 100.0 + 300.0 * r(h, w, 1), z, z], -1),
              "transmission": r(h, w), "sheen_albedo_scaling_lut": 0.5 * r(32, 32), "preintegrated_charlie": 0.3 * r(32, 32)}
    planes = {k: v.contiguous() for k, v in planes.items()}
    px = w * h
    base = 16 * 3 + 4 + 16 * 2
    extra = {1: 32, 2: 16, 4: 32, 8: 16, 16: 4}
    print(f"{w}x{h}, {sa.LightCount} lights; us per call (HIP events, {a.steps} calls), GB/s by the bytes of the planes the permutation reads and writes")
    t0 = timed(lambda: api.pbr_shade(ctx, g, f["camera"], sa, ibl), a.steps)
    print(f"  default shade (pbr_shade_kernel)        : {t0:8.1f} us  {px * base / t0 / 1e3:7.0f} GB/s of {base} B/px")
    for name, flags in (("clear coat", 1), ("sheen", 2), ("anisotropy", 4), ("iridescence", 8), ("transmission", 16), ("all five", 31)):
        b = base + sum(v for k, v in extra.items() if flags & k)
        tt = timed(lambda: api.pbr_shade_layers(ctx, g, planes, flags, f["camera"], sa, ibl, iridescence_ior=1.33, anisotropy_rotation=0.7), a.steps)
        print(f"  layers: {name:<12} (flags {flags:2d})         : {tt:8.1f} us  {px * b / tt / 1e3:7.0f} GB/s of {b} B/px")
    import chain_util

    ssa = chain_util.shadowed_shade_attribs(len(ibl.pre) - 1)
    slices, infos = chain_util.make_shadow_inputs()
    sm = torch.from_numpy(np.stack(slices)).to(dev)
    b = base + sum(extra.values())
    tt = timed(lambda: api.pbr_shade_layers(ctx, g, planes, 31, f["camera"], ssa, ibl, iridescence_ior=1.33, anisotropy_rotation=0.7, shadows=(sm, infos, 3)), a.steps)
    ts = timed(lambda: api.pbr_shade(ctx, g, f["camera"], ssa, ibl, shadows=(sm, infos, 3)), a.steps)
    print(f"  all five + 2 shadow maps, {ssa.LightCount} lights      : {tt:8.1f} us  {px * b / tt / 1e3:7.0f} GB/s of {b} B/px   (default shade with the same shadows: {ts:.1f} us)")


if __name__ == "__main__":
    main()
