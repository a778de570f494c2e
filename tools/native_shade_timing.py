#!/usr/bin/env python3
"""mifx_pbr_shade_execute_native on a 3840x2160 G-buffer in Hydrogent's texture formats: the per-format instance of the kernel (MIFX_NATIVE_SHADE_PER_FORMAT=1) against the generic instance
(run-time format switch per load; the default) and against the fp32-contract shade.  HIP-event time per call, cube aprons included.

    python tools/native_shade_timing.py [--width 3840 --height 2160 --steps 30]"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import api, synth  # noqa: E402


def timed(fn, steps):
    for _ in range(5):
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
    p.add_argument("--steps", type=int, default=30)
    a = p.parse_args()
    w, h = a.width, a.height
    ctx = api.PostFXContext(0)
    env = synth.make_sky_cube(64, ctx.device)
    ibl = api.precompute_ibl(ctx, env, lut_size=128, irradiance_size=16, prefiltered_size=128, lut_samples=128, diffuse_samples=256, specular_samples=64)
    sa = synth.make_lights()
    sa.PrefilteredCubeLastMip = float(len(ibl.pre) - 1)
    f = synth.make_frame(synth.Scene(), 3, w, h, ctx.device)
    names = ("base_color", "normal", "material", "depth")
    native = {k: (api.image_export(ctx, f[k], api.HYDROGENT_GBUFFER_FORMATS[k]), api.HYDROGENT_GBUFFER_FORMATS[k]) for k in names}
    g32 = {k: f[k] for k in names}
    t_generic = timed(lambda: api.pbr_shade_native(ctx, native, w, f["camera"], sa, ibl), a.steps)
    os.environ["MIFX_NATIVE_SHADE_PER_FORMAT"] = "1"
    t_fixed = timed(lambda: api.pbr_shade_native(ctx, native, w, f["camera"], sa, ibl), a.steps)
    del os.environ["MIFX_NATIVE_SHADE_PER_FORMAT"]
    t_fp32 = timed(lambda: api.pbr_shade(ctx, g32, f["camera"], sa, ibl), a.steps)
    px = w * h
    nat_bytes = px * (4 + 8 + 2 + 4 + 8 + 8)   # RGBA8 + RGBA16F + RG8 + R32F in, 2 x RGBA16F out
    f32_bytes = px * (16 * 3 + 4 + 16 * 2)
    print(f"{w}x{h} shade (+ aprons; outputs allocated per call by the Python wrapper for the native route):")
    print(f"  native G-buffer, per-format instance : {t_fixed:8.1f} us   ({nat_bytes / t_fixed / 1e6:6.2f} TB/s of {nat_bytes / px} B/px)")
    print(f"  native G-buffer, generic instance    : {t_generic:8.1f} us")
    print(f"  fp32 planes (the contract)           : {t_fp32:8.1f} us   ({f32_bytes / t_fp32 / 1e6:6.2f} TB/s of {f32_bytes / px} B/px)")


if __name__ == "__main__":
    main()
