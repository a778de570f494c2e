#!/usr/bin/env python3
"""Would a HIP graph help?  Captures one frame of the chain (all its launches, both streams) with torch.cuda.graph and compares replaying it with issuing the
same frame through the launch stream.  The replay repeats one frame index (same pointers, same work), so this is a timing probe, not a product path.

    python tools/graph_probe.py [--width 3840 --height 2160 --steps 40]"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import tiling  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--width", type=int, default=3840)
    p.add_argument("--height", type=int, default=2160)
    p.add_argument("--steps", type=int, default=40)
    a = p.parse_args()
    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    r = tiling.TiledChain(0, tables["sobol_256d"], tables["scrambling_tile"], 0, 1, a.width, a.height)
    r.build_inputs(n_frames=4)
    for _ in range(10):
        r.step()
    torch.cuda.synchronize()

    def timed(fn):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            fn()
        issue = (time.perf_counter() - t0) / a.steps * 1e3
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3, issue

    stream_ms, stream_issue = timed(r.step)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):
            r.step()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        r.step()
    torch.cuda.synchronize()
    graph_ms, graph_issue = timed(g.replay)
    print(f"{a.width}x{a.height}: stream launches {stream_ms:.4f} ms/frame (host issue {stream_issue:.4f} ms), graph replay {graph_ms:.4f} ms/frame (host issue {graph_issue:.4f} ms)")


if __name__ == "__main__":
    main()
