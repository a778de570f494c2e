#!/usr/bin/env python3
"""Hashes of what the chain produces for a few frames (final image, SSAO / SSR outputs, TAA history), for comparing run-time variants of one build that must be
bit-identical:   MIFX_A3_WINDOW=1 python tools/variant_hash.py > a; MIFX_A3_WINDOW=0 python tools/variant_hash.py > b; diff a b"""
import argparse
import hashlib
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import tiling  # noqa: E402


def h(t):
    return hashlib.sha1(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()[:16]


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--frames", type=int, default=4)
    a = p.parse_args()
    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    r = tiling.TiledChain(0, tables["sobol_256d"], tables["scrambling_tile"], 0, 1, a.width, a.height)
    r.build_inputs(n_frames=a.frames)
    for i in range(a.frames):
        r.step(i)
        torch.cuda.synchronize()
        print(i, "ldr", h(r.out), "ssao", h(r.chain.effect_output("ssao")), "ssr", h(r.chain.effect_output("ssr")))


if __name__ == "__main__":
    main()
