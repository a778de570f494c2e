#!/usr/bin/env python3
"""Headroom check: two independent 4K chains on two streams of one GPU vs one chain -- how much idle issue capacity does a single chain leave?"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import tiling  # noqa: E402

tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
W, H, K = 3840, 2160, 30
runners = [tiling.TiledChain(0, tables["sobol_256d"], tables["scrambling_tile"], r, 2, W, H) for r in range(2)]
for r in runners:
    r.build_inputs()
streams = [torch.cuda.Stream(), torch.cuda.Stream()]


def run(active, steps, first):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        for j in active:
            with torch.cuda.stream(streams[j]):
                runners[j].step(first + i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


run([0, 1], 20, 0)
one = run([0], K, 20)
two = run([0, 1], K, 20 + K)
print(f"one chain: {one:.3f} ms/frame; two concurrent chains: {two:.3f} ms per pair = {two / 2:.3f} ms/frame ({one / (two / 2):.3f}x)")
