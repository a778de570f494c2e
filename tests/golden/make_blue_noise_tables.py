#!/usr/bin/env python3
"""Extracts the blue-noise sampler tables (Sobol_256d[256], ScramblingTile[128*128*8]) from the reference's
PostProcess/Common/src/SamplerBlueNoiseErrorDistribution_128x128_OptimizedFor_2d2d2d2d_1spp.cpp into a binary
fixture.  In a real integration the adapter passes the reference's arrays to mifx_postfx_create(); tests and
bench.py (which cannot read /root/reference on the GPU box) use this fixture instead.
Run in the build container:  python tests/golden/make_blue_noise_tables.py"""
import os
import re
import sys

import numpy as np

REF = os.environ.get("MIFX_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "PostProcess/Common/src/SamplerBlueNoiseErrorDistribution_128x128_OptimizedFor_2d2d2d2d_1spp.cpp")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "blue_noise_tables.npz")


def grab(text, name):
    m = re.search(name + r"\s*\[[^\]]*\]\s*=\s*\{(.*?)\};", text, re.S)
    assert m, name
    vals = [int(v) for v in re.findall(r"\d+", m.group(1))]
    return np.array(vals, dtype=np.uint8)


def main():
    text = open(SRC).read()
    sobol = grab(text, "Sobol_256d")
    tile = grab(text, "ScramblingTile")
    assert sobol.size == 256 and tile.size == 128 * 128 * 8, (sobol.size, tile.size)
    assert sobol[0] == 32 and tile[0] == 162
    np.savez_compressed(OUT, sobol_256d=sobol, scrambling_tile=tile)
    print(OUT, os.path.getsize(OUT))


if __name__ == "__main__":
    sys.exit(main())
