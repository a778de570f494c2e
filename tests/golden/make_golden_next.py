#!/usr/bin/env python3
"""Generates tests/golden/next_golden.npz: outputs of the REFERENCE (oracle/_ref/libmifx_ref.so, the reference's shader source compiled for the CPU)
for the SURVEY 8f rows built after the chain: depth of field (temporal + Karis, three frames), the environment-map background, half-resolution
SSAO and SSR, and the reversed-depth SSR / SSAO outputs.  The fixture carries its own inputs.
Run in the build container (needs /root/reference for oracle/_ref):   python tests/golden/make_golden_next.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)

W, H, FRAMES = 80, 56, 3
LENS = (12.0, 1.2, 135.0)


def frames_for(reversed_depth):
    import torch

    from diligentfx_amd import synth

    scene = synth.Scene()
    out = []
    for fi in range(4, 4 + FRAMES):
        f = synth.make_frame(scene, fi, W, H, torch.device("cpu"), reversed_depth=reversed_depth)
        f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = LENS
        g = {k: v.numpy() for k, v in f.items() if isinstance(v, torch.Tensor)}
        g["color"] = np.ascontiguousarray(np.concatenate([g["base_color"][..., :3] * 2.0 + 0.1 * np.abs(g["normal"][..., :3]), g["base_color"][..., 3:4] * 0.5 + 0.25], -1).astype(np.float32))
        out.append((fi, g, bytes(f["camera"]), bytes(f["prev_camera"])))
    return out


def run(lib, prefix, reversed_depth, data=None):
    """Runs every row on the frames (generated, or taken from `data`); returns {name: array}."""
    import chain_util
    import cpu_chain
    from diligentfx_amd import binding as B, synth
    from util import blue_noise_tables, tone_mapping_attribs_bytes
    import torch

    tag = "rev" if reversed_depth else "fwd"
    res = {}
    if data is None:
        frames = frames_for(reversed_depth)
        for i, (fi, g, cam, prev) in enumerate(frames):
            for k in ("depth", "prev_depth", "normal", "material", "motion", "color"):
                res[f"{tag}{i}_in_{k}"] = g[k]
            res[f"{tag}{i}_camera"], res[f"{tag}{i}_prev_camera"] = np.frombuffer(cam, np.uint8), np.frombuffer(prev, np.uint8)
            res[f"{tag}{i}_index"] = np.array([fi])
    else:
        frames = []
        for i in range(FRAMES):
            g = {k: data[f"{tag}{i}_in_{k}"] for k in ("depth", "prev_depth", "normal", "material", "motion", "color")}
            frames.append((int(data[f"{tag}{i}_index"][0]), g, data[f"{tag}{i}_camera"].tobytes(), data[f"{tag}{i}_prev_camera"].tobytes()))
    tables = blue_noise_tables()
    chain = cpu_chain.CpuChain(lib, prefix, reversed_depth=reversed_depth)
    half = cpu_chain.CpuChain(lib, prefix)
    dofc = cpu_chain.CpuChain(lib, prefix)
    attribs = B.DOFAttribs.default()
    attribs.MaxCircleOfConfusion = 0.02
    for i, (fi, g, cam, prev) in enumerate(frames):
        pf = chain.postfx(fi, g["depth"], g["prev_depth"], g["motion"], cam, prev, tables)
        res[f"{tag}{i}_out_ssao"] = chain.ssao(pf, g["depth"], g["normal"], B.SSAOAttribs.default())
        res[f"{tag}{i}_out_ssr"] = chain.ssr(pf, g["color"], g["depth"], g["normal"], g["material"], g["motion"], B.SSRAttribs.default())
        if not reversed_depth:
            pfh = half.postfx(fi, g["depth"], g["prev_depth"], g["motion"], cam, prev, tables)
            res[f"{tag}{i}_out_ssao_half"] = half.ssao(pfh, g["depth"], g["normal"], B.SSAOAttribs.default(), half_resolution=True)
            res[f"{tag}{i}_out_ssr_half"] = half.ssr(pfh, g["color"], g["depth"], g["normal"], g["material"], g["motion"], B.SSRAttribs.default(), half_resolution=True)
            res[f"{tag}{i}_out_dof"] = dofc.dof({"frame": fi, "cam": cam, "closest_motion": pf["closest_motion"]}, g["color"], g["depth"], attribs, flags=3)
    if not reversed_depth:
        fi, g, cam, prev = frames[-1]
        if data is None:
            env = chain_util.box_mips(np.minimum(synth.make_sky_cube(16, torch.device("cpu")).numpy(), 500.0).astype(np.float32))
            for m, e in enumerate(env):
                res[f"env{m}"] = e
        else:
            env = [data[f"env{m}"] for m in range(5)]
        color, motion = np.full((H, W, 4), -7.0, np.float32), np.full((H, W, 2), -7.0, np.float32)
        name = "ref_envmap" if prefix == "ref_" else "oracle_envmap"
        lib.call(name, [env, g["depth"]], [color, motion], cam0=cam, cam1=prev, attribs=tone_mapping_attribs_bytes(0), fval=[0.3, 1.0, 0.0, 1.5, 1.0, 0.75], ival=[0, 1])
        res["out_envmap_color"], res["out_envmap_motion"] = color, motion
    return res


def main():
    import pyref

    ref = pyref.ref_lib()
    assert ref is not None, "build oracle/_ref first (python oracle/build.py)"
    data = {}
    data.update(run(ref, "ref_", False))
    data.update(run(ref, "ref_", True))
    out = os.path.join(HERE, "next_golden.npz")
    np.savez_compressed(out, **data)
    print(out, os.path.getsize(out))


if __name__ == "__main__":
    main()
