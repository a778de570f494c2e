#!/usr/bin/env python3
"""Why does the PBR shade take 0.21 ms per launch as BASELINE configs[2] (the shade alone, back to back) and 0.17 ms inside the chain?  Same kernel, same G-buffers.
The shade alone back to back / with the device idle for 2 ms between launches / with a streaming copy of the frame's size class between launches; HIP events around the kernel."""
import ctypes
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import binding as B, tiling  # noqa: E402

tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
r = tiling.StageRunner("pbr", 0, tables["sobol_256d"], tables["scrambling_tile"], 3840, 2160)
r.build_inputs(n_frames=8)
n = 3840 * 2160 * 4 * 4
src, dst = torch.ones(n, dtype=torch.float32, device=r.dev), torch.empty(n, dtype=torch.float32, device=r.dev)
ctx = r.chain.postfx


def run(mode, frames=40):
    for _ in range(12):
        r.step()
    r.arm_kernel_timing("pbr_shade_kernel", frames)
    a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(frames):
        r.step()
        if mode == "idle":
            torch.cuda.synchronize()
            time.sleep(0.002)
        elif mode == "copy":
            B.check(ctx.lib.mifx_debug_stream_copy(ctx.handle, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_uint64(4 * n)))
    z.record()
    torch.cuda.synchronize()
    t = sorted(r.kernel_times_ms(frames))
    r.arm_kernel_timing(None, 0)
    return t[len(t) // 2], t[0], t[-1], a.elapsed_time(z) / frames


for mode in ("back to back", "idle", "copy", "back to back"):
    med, lo, hi, per = run(mode)
    print(f"shade at 3840x2160, {mode:13s}: kernel median {med * 1e3:.1f} us (min {lo * 1e3:.1f}, max {hi * 1e3:.1f}); {per:.3f} ms per step")
