#!/usr/bin/env python3
"""Generates tests/golden/chain_golden.npz: inputs and outputs of the REFERENCE running the full chain on the CPU
(oracle/_ref/libmifx_ref.so = the reference's own shader source compiled through oracle/ref/hlsl_shim.h).

The fixture carries its own inputs (G-buffers, cameras, IBL maps) so that nothing has to be regenerated bit-identically elsewhere:
  3 consecutive frames at 96x64; per frame the final LDR image and the effect outputs (SSAO, SSR, TAA, Bloom, radiance, composite).
Run in the build container (needs /root/reference for oracle/_ref):   python tests/golden/make_golden_chain.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

W, H, FRAMES = 96, 64, 3


def main():
    import torch

    import chain_util
    import cpu_chain
    import pyref
    from diligentfx_amd import synth

    ref = pyref.ref_lib()
    assert ref is not None, "build oracle/_ref first (python oracle/build.py)"
    ibl = chain_util.make_ibl(ref, "ref_")
    chain = cpu_chain.CpuChain(ref, "ref_")
    scene = synth.Scene()
    data = {"lut": ibl["lut"], "irradiance": ibl["irradiance"][0]}
    for i, m in enumerate(ibl["prefiltered"]):
        data[f"prefiltered{i}"] = m
    for frame in range(FRAMES):
        keep = {}
        final = chain_util.run_frame(chain, scene, frame, W, H, ibl, keep)
        g = keep["gbuffer"]
        for k in ("depth", "normal", "base_color", "material", "motion", "prev_depth"):
            data[f"f{frame}_in_{k}"] = g[k]
        data[f"f{frame}_camera"] = np.frombuffer(keep["camera"], np.uint8)
        data[f"f{frame}_prev_camera"] = np.frombuffer(keep["prev_camera"], np.uint8)
        for k in ("radiance", "specular_ibl", "ssao_out", "ssr_out", "composite", "taa_out", "bloom_out"):
            data[f"f{frame}_out_{k}"] = keep[k]
        data[f"f{frame}_out_final"] = final
    data["shade_attribs"] = np.frombuffer(bytes(chain_util.shade_attribs(len(ibl["prefiltered"]) - 1)), np.uint8)
    out = os.path.join(HERE, "chain_golden.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
