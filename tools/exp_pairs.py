#!/usr/bin/env python3
"""Which kinds of work share the GPU for free?  Three kernels' worth of work on their own streams, alone and in pairs, at 3840x2160:
   S = PostFX prep + SSAO A2..A8 (gathers: the vector L1's tag look-ups and vector ALU; 17 % of the copy rate in its largest kernel)
   P = the PBR shade (vector ALU, 71 % of the copy rate)
   C = a device-to-device copy of the frame's size class (pure HBM streaming)
Prints the time of each alone, of each pair side by side, and the share of the shorter one that disappeared ("hidden").  If gather work hid under streaming work,
a frame pipelined by resource class would approach its HBM time; round 4's three-lane mode (mifx_chain_set_overlap 3) put the two gather kernels side by side and
conserved their total."""
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
W, H, K = 3840, 2160, 40
ssao = tiling.StageRunner("ssao", 0, tables["sobol_256d"], tables["scrambling_tile"], W, H)
pbr = tiling.StageRunner("pbr", 0, tables["sobol_256d"], tables["scrambling_tile"], W, H)
for r in (ssao, pbr):
    r.build_inputs(n_frames=6)
dev = ssao.dev
n = W * H * 4 * 4  # floats: four float4 planes read, four written per step (~1 GB of traffic, ~0.18 ms)
src, dst = torch.ones(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
copy_ctx = tiling.StageRunner("pbr", 0, tables["sobol_256d"], tables["scrambling_tile"], 64, 64).chain.postfx
lib = copy_ctx.lib
streams = {k: torch.cuda.Stream() for k in "SPC"}


def step(k, i):
    with torch.cuda.stream(streams[k]):
        if k == "S":
            ssao.ctx.sync_stream()
            ssao.step(i)
        elif k == "P":
            pbr.ctx.sync_stream()
            pbr.step(i)
        else:
            copy_ctx.sync_stream()
            B.check(lib.mifx_debug_stream_copy(copy_ctx.handle, ctypes.c_void_p(src.data_ptr()), ctypes.c_void_p(dst.data_ptr()), ctypes.c_uint64(4 * n)))


def run(active, steps):
    for i in range(8):
        for k in active:
            step(k, i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for k in active:
            step(k, 8 + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


alone = {k: run([k], K) for k in "SPC"}
print("alone (ms per step): " + ", ".join(f"{k} {v:.3f}" for k, v in alone.items()))
for a, b in (("S", "C"), ("P", "C"), ("S", "P"), ("S", "S2")):
    if b == "S2":
        continue
    t = run([a, b], K)
    hidden = (alone[a] + alone[b] - t) / min(alone[a], alone[b])
    print(f"{a} beside {b}: {t:.3f} ms per pair of steps; sum alone {alone[a] + alone[b]:.3f}; hidden {hidden * 100:.0f} % of the shorter")
