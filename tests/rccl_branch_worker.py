"""One rank of tests/test_gpu_sharded.py::test_rccl_branch_with_several_processes: a process of its own that joins an N-rank communicator through the C ABI
(mifx_comm_create = ncclCommInitRank), runs mifx_chain_execute_sharded for a few frames and compares its band of every frame and its history planes on band + halo with
the unsharded chain it runs beside it.  The RCCL the library loads is MIFX_RCCL_PATH (the stand-in of tests/fake_rccl when all ranks share one GPU).

    python tests/rccl_branch_worker.py <rank> <world> <id file> <width> <height> <frames> [frames | selftest | selftest_absent | selftest_mismatch]
(the selftest modes run mifx_comm_self_test only: all ranks; the last rank never calling it; rank 0 announcing another slab size than the others)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from diligentfx_amd import api, synth, tiling  # noqa: E402
from diligentfx_amd.sharded import HISTORY_PLANES  # noqa: E402


def main():
    rank, world, idfile, w, h, frames = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
    mode = sys.argv[7] if len(sys.argv) > 7 else "frames"
    tables = np.load(os.path.join(ROOT, "tests", "golden", "blue_noise_tables.npz"))
    sobol, tile = tables["sobol_256d"], tables["scrambling_tile"]
    chain, ref = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    dev = chain.device
    if rank == 0:
        uid = api.Comm.unique_id()
        with open(idfile + ".tmp", "wb") as f:
            f.write(uid)
        os.replace(idfile + ".tmp", idfile)
    else:
        t0 = time.time()
        while not os.path.exists(idfile):
            if time.time() - t0 > 60:
                raise SystemExit("no unique id from rank 0")
            time.sleep(0.05)
        uid = open(idfile, "rb").read()
    comm = api.Comm.create(chain.postfx, uid, rank, world)
    r, n, is_rccl = comm.info()
    assert (r, n) == (rank, world) and is_rccl, (r, n, is_rccl)
    if mode != "frames":
        from diligentfx_amd import binding as B

        if mode == "selftest_absent" and rank == world - 1:
            time.sleep(float(os.environ.get("MIFX_FAKE_RCCL_TIMEOUT", "2")) + 1.0)  # never posts; leaves after the others have given up
            print(f"rank {rank}/{world}: stayed away from the self test", flush=True)
        else:
            try:
                comm.self_test(chain.postfx, 4096 if (mode == "selftest_mismatch" and rank == 0) else 8192, timeout_ms=20000)
                print(f"rank {rank}/{world}: self test OK", flush=True)
            except B.MifxError as e:
                print(f"rank {rank}/{world}: self test FAILED: {e}", flush=True)
        comm.close()
        chain.close()
        ref.close()
        return 0
    env = synth.make_sky_cube(32, dev).clamp(max=200.0)
    ibl = api.precompute_ibl(ref.postfx, env, lut_size=32, irradiance_size=8, prefiltered_size=32, lut_samples=32, diffuse_samples=32, specular_samples=16)
    sa = synth.make_lights()
    sa.PrefilteredCubeLastMip = float(len(ibl.pre) - 1)
    scene = synth.Scene()
    fs = [synth.make_frame(scene, 16 + i, w, h, dev) for i in range(frames)]
    max_motion = int(max(float(f["motion"][..., 1].abs().max()) for f in fs) * 0.5 * h) + 2
    cuts = list(tiling.band_cuts(fs[0], chain.ssr_attribs, world, min(96, h // world)))
    chain.set_sharding(comm, cuts, max_motion)
    chain.set_overlap(int(os.environ.get("MIFX_SHARD_OVERLAP", "3")))  # the sharded frame as lanes across frames, as bench.py --gpus N runs it (3; the test also runs 2)
    b, e = cuts[rank], cuts[rank + 1]
    out, want = torch.full((h, w, 4), -1.0, device=dev), torch.zeros(h, w, 4, device=dev)
    bad = []
    for i, f in enumerate(fs):
        ref.execute(ref.bind_frame(16 + i, f, ibl, sa, want))
        bound = chain.bind_frame(16 + i, f, ibl, sa, out)
        chain.execute_sharded(bound)
        torch.cuda.synchronize()
        if not torch.equal(out[b:e], want[b:e]):
            bad.append(f"frame {i}: {int((out[b:e] != want[b:e]).any(dim=-1).sum())} pixels of the band differ")
        # the last level of SSAO's depth pyramid, which the ranks reduce in pieces and all-gather (round 6): whole and equal to the unsharded chain's on every rank
        l4, l4ref = chain.effect("ssao").get_intermediate("prefiltered_depth4"), ref.effect("ssao").get_intermediate("prefiltered_depth4")
        if not torch.equal(l4, l4ref):
            rows = sorted(set(torch.nonzero((l4 != l4ref).any(dim=-1)).flatten().tolist()))
            bad.append(f"frame {i}: level 4 of the prefiltered depth differs on rows {rows[:12]}{'...' if len(rows) > 12 else ''} of {l4.shape[0]}")
        info = chain.shard_info(bound)
        for name, field in HISTORY_PLANES:
            halo = getattr(info, field)
            got, exp = chain.shard_plane(name), ref.shard_plane(name)
            lo, hi = max(b - halo, 0), min(e + halo, h)
            cols = w * (4 if name in ("taa_history", "ssr_history_radiance") else 1)
            if not torch.equal(got[lo:hi, :cols], exp[lo:hi, :cols]):
                bad.append(f"frame {i}: history plane {name} differs on band + halo")
    print(f"rank {rank}/{world}: band [{b}, {e}) of {w}x{h}, {frames} frames, RCCL branch (is_rccl {int(is_rccl)}): " + ("bit-identical to the unsharded chain" if not bad else "; ".join(bad)), flush=True)
    comm.close()
    chain.close()
    ref.close()
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
