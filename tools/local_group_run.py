#!/usr/bin/env python3
"""N ranks of the sharded chain on ONE GPU, exchanges below the C ABI: mifx_chain_execute_sharded with an in-process group (mifx_comm_create_local_group: one host
thread and one stream per rank, device copies with event hand-shakes where RCCL would send / receive), at the size and with the cost-weighted cuts bench.py --gpus N uses.
Every rank's band of every frame and its history planes on band + halo are compared with the unsharded chain bit for bit.

    python tools/local_group_run.py [--world 8 --width 7680 --height 4320 --frames 4]"""
import argparse
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diligentfx_amd import api, synth, tiling  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--world", type=int, default=8)
    p.add_argument("--width", type=int, default=7680)
    p.add_argument("--height", type=int, default=4320)
    p.add_argument("--frames", type=int, default=4)
    p.add_argument("--dof", action="store_true", help="depth of field (temporal smoothing + Karis weights) in every chain")
    p.add_argument("--half", action="store_true", help="half-resolution SSAO and SSR")
    p.add_argument("--auto-exposure", action="store_true")
    a = p.parse_args()
    w, h, world = a.width, a.height, a.world
    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    sobol, tile = tables["sobol_256d"], tables["scrambling_tile"]
    ref = api.Chain(0, sobol, tile)
    dev = ref.device
    env = synth.make_sky_cube(64, dev)
    ibl = api.precompute_ibl(ref.postfx, env, lut_size=128, irradiance_size=16, prefiltered_size=64, lut_samples=128, diffuse_samples=256, specular_samples=64)
    sa = synth.make_lights()
    sa.PrefilteredCubeLastMip = float(len(ibl.pre) - 1)
    scene = synth.Scene()
    frames = [synth.make_frame(scene, i, w, h, dev) for i in range(a.frames)]
    max_motion = int(max(float(f["motion"][..., 1].abs().max()) for f in frames) * 0.5 * h) + 2
    cuts = list(tiling.band_cuts(frames[0], ref.ssr_attribs, world, min(192, h // world)))
    print(f"{w}x{h}, {world} ranks in one process on one GPU; cuts {cuts}; max motion {max_motion} rows")
    chains = [api.Chain(0, sobol, tile) for _ in range(world)]
    for c in chains + [ref]:
        if a.half:
            c.set_effect_feature_flags(ssao_feature_flags=2, ssr_feature_flags=2)
        if a.auto_exposure:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
        if a.dof:
            from diligentfx_amd import binding as B

            da = B.DOFAttribs.default()
            c.set_depth_of_field(da, 3)
    if a.dof:
        for f in frames:
            f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = 12.0, 1.2, 135.0
    print("options:", ", ".join(n for n, on in (("depth of field", a.dof), ("half-resolution SSAO + SSR", a.half), ("auto exposure", a.auto_exposure)) if on) or "none")
    comms = api.Comm.local_group(chains[0].postfx, world)
    outs = [torch.zeros(h, w, 4, device=dev) for _ in range(world)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(world)]
    for r in range(world):
        chains[r].set_sharding(comms[r], cuts, max_motion)
    want = torch.zeros(h, w, 4, device=dev)
    bad_bands = bad_hist = 0
    for i, f in enumerate(frames):
        ref.execute(ref.bind_frame(i, f, ibl, sa, want))
        torch.cuda.synchronize()
        errors = []

        def run(r):
            try:
                with torch.cuda.stream(streams[r]):
                    chains[r].execute_sharded(chains[r].bind_frame(i, f, ibl, sa, outs[r]))
                streams[r].synchronize()
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))

        t0 = time.perf_counter()
        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3
        if errors:
            sys.exit(f"frame {i}: {errors}")
        for r in range(world):
            bad_bands += int(not torch.equal(outs[r][cuts[r]:cuts[r + 1]], want[cuts[r]:cuts[r + 1]]))
        for name in ("taa_history", "ssr_history_radiance", "ssr_history_variance", "ssao_history_ao", "ssao_history_len"):
            full = ref.shard_plane(name)
            for r in range(world):
                lo, hi = max(cuts[r] - 8, 0), min(cuts[r + 1] + 8, h)
                bad_hist += int(not torch.equal(chains[r].shard_plane(name)[lo:hi], full[lo:hi]))
        print(f"  frame {i}: all {world} ranks done in {ms:.1f} ms of wall time (one GPU shared); bands differing so far {bad_bands}, history planes differing so far {bad_hist}")
    print(f"RESULT: {a.frames} frames x {world} ranks: {bad_bands} bands and {bad_hist} history planes (band + 8 halo rows) differ from the unsharded chain")
    for r in range(world):
        chains[r].set_sharding(None)
        comms[r].close()
        chains[r].close()
    ref.close()
    sys.exit(1 if bad_bands or bad_hist else 0)


if __name__ == "__main__":
    main()
