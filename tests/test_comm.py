"""The sharded chain with the exchanges inside the library (mifx_comm_*, mifx_chain_execute_sharded).

CPU: entry points exist, arguments are checked, a missing / unusable RCCL is MIFX_ERR_COMM, never a crash.
GPU: (i) an RCCL communicator of one rank on the device (the API calls work, the frame equals mifx_chain_execute's);
     (ii) 2, 3 and 4 ranks as an in-process group on one GPU, one host thread per rank -- the exchange code of mifx_chain_execute_sharded with device
          copies in place of ncclSend / ncclRecv (RCCL refuses two ranks on one device): every rank's band of every frame, and the history planes on
          band + halo, must equal the unsharded chain bit for bit."""
import ctypes
import threading

import numpy as np
import pytest

from util import blue_noise_tables


def test_comm_entry_points_check_their_arguments(mifx_lib):
    from diligentfx_amd import binding as B

    for name in ("mifx_comm_get_unique_id", "mifx_comm_create", "mifx_comm_create_local_group", "mifx_comm_destroy", "mifx_comm_get_info", "mifx_chain_set_sharding",
                 "mifx_chain_execute_sharded"):
        assert hasattr(mifx_lib, name)
    assert mifx_lib.mifx_comm_get_unique_id(None) == -1                                    # MIFX_ERR_INVALID_ARG
    h = ctypes.c_void_p()
    buf = (ctypes.c_uint8 * 128)()
    assert mifx_lib.mifx_comm_create(None, buf, 0, 1, ctypes.byref(h)) == -1
    assert mifx_lib.mifx_chain_execute_sharded(None, None, None) == -1
    mifx_lib.mifx_comm_destroy(None)                                                          # like free(NULL)
    # without a device RCCL cannot hand out an id: the call reports MIFX_ERR_COMM (or succeeds where RCCL works without one); it never crashes
    st = mifx_lib.mifx_comm_get_unique_id(buf)
    assert st in (0, -6), st
    if st == -6:
        assert mifx_lib.mifx_status_string(st) == b"MIFX_ERR_COMM" and len(mifx_lib.mifx_last_error()) > 0
    assert B.ShardInfo is not None


def _frames(chain, scene, n, w, h):
    from diligentfx_amd import synth

    return [synth.make_frame(scene, i, w, h, chain.device) for i in range(n)]


def _setup(w, h):
    import torch

    from diligentfx_amd import api, synth

    sobol, tile = blue_noise_tables()
    ref = api.Chain(0, sobol, tile)
    env = synth.make_sky_cube(32, ref.device)
    ibl = api.precompute_ibl(ref.postfx, env, lut_size=64, irradiance_size=8, prefiltered_size=32, lut_samples=64, diffuse_samples=128, specular_samples=32)
    sa = synth.make_lights()
    sa.PrefilteredCubeLastMip = float(len(ibl.pre) - 1)
    return ref, ibl, sa, synth.Scene(), (sobol, tile), torch


@pytest.mark.gpu
def test_rccl_communicator_of_one_rank(mifx_lib):
    from diligentfx_amd import api

    w, h = 320, 192
    ref, ibl, sa, scene, (sobol, tile), torch = _setup(w, h)
    chain = api.Chain(0, sobol, tile)
    comm = api.Comm.create(chain.postfx, api.Comm.unique_id(), 0, 1)
    assert comm.info() == (0, 1, True)
    chain.set_sharding(comm, [0, h], 8)
    a, b = torch.zeros(h, w, 4, device=ref.device), torch.zeros(h, w, 4, device=ref.device)
    for i, f in enumerate(_frames(ref, scene, 3, w, h)):
        ref.execute(ref.bind_frame(i, f, ibl, sa, a))
        chain.execute_sharded(chain.bind_frame(i, f, ibl, sa, b))
        assert torch.equal(a, b)
    chain.set_sharding(None)
    comm.close()
    chain.close()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,size,cuts,mode", [(2, (384, 512), None, ""), (3, (320, 640), (0, 200, 430, 640), ""), (4, (256, 1024), None, ""),
                                                  (3, (320, 640), (0, 200, 430, 640), "auto exposure"), (2, (384, 512), None, "half resolution"),
                                                  (4, (320, 640), (0, 280, 304, 330, 640), "thin bands"), (3, (320, 640), (0, 200, 430, 640), "depth of field"),
                                                  (3, (320, 640), (0, 200, 430, 640), "two lanes"), (3, (320, 640), (0, 200, 430, 640), "two lanes + depth of field + auto exposure"),
                                                  (3, (320, 640), (0, 200, 430, 640), "three lanes"), (4, (320, 640), (0, 280, 304, 330, 640), "three lanes + depth of field + auto exposure")])  # (halos taller than a band: rows from the rank beyond the neighbour)
def test_sharded_execute_in_process_group(mifx_lib, world, size, cuts, mode):
    """mode: auto exposure = the luminance rows travel after phase 3 and phase 4 follows; half resolution = SSAO and SSR with FEATURE_FLAG_HALF_RESOLUTION; two lanes =
    mifx_chain_set_overlap 2 on every rank's chain (phases 0 - 2 on a side stream, phase 3 beside the next frame's first phases; the frames are queued without a
    synchronisation in between)."""
    from diligentfx_amd import api

    w, h = size
    cuts = list(cuts) if cuts else [h * r // world for r in range(world + 1)]
    ref, ibl, sa, scene, (sobol, tile), torch = _setup(w, h)
    frames = _frames(ref, scene, 5, w, h)
    if "depth of field" in mode:
        for fr in frames:
            fr["camera"].fFocusDistance, fr["camera"].fFStop, fr["camera"].fFocalLength = 12.0, 1.2, 135.0
    max_motion = int(max(float(f["motion"][..., 1].abs().max()) for f in frames) * 0.5 * h) + 2
    chains = [api.Chain(0, sobol, tile) for _ in range(world)]
    for c in chains + [ref]:
        if "auto exposure" in mode:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
        if mode == "half resolution":
            c.set_effect_feature_flags(ssao_feature_flags=2, ssr_feature_flags=2)
        if "depth of field" in mode:
            from diligentfx_amd import binding as B

            da = B.DOFAttribs.default()
            da.MaxCircleOfConfusion = 0.02
            c.set_depth_of_field(da, 3)
    comms = api.Comm.local_group(chains[0].postfx, world)
    outs = [torch.zeros(h, w, 4, device=ref.device) for _ in range(world)]
    streams = [torch.cuda.Stream(device=ref.device) for _ in range(world)]
    for r in range(world):
        assert comms[r].info() == (r, world, False)
        chains[r].set_sharding(comms[r], cuts, max_motion)
        if "two lanes" in mode:
            chains[r].set_overlap(2)
        if "three lanes" in mode:  # prep + SSAO on a lane of their own beside the shade and SSR (api_comm.cpp execute_sharded_impl)
            chains[r].set_overlap(3)
    want = torch.zeros(h, w, 4, device=ref.device)
    errors = []
    before = [c.stats() for c in comms]
    assert all(st["ranks_in_communicator"] == world and st["world"] == world and not st["is_rccl"] for st in before)
    for i, f in enumerate(frames):
        ref.execute(ref.bind_frame(i, f, ibl, sa, want))
        torch.cuda.synchronize()
        for c in comms:
            c.set_timing(i == len(frames) - 1)  # (mifx_comm_set_timing: the exchange groups of the last frame between HIP events)

        def run(r):
            try:
                with torch.cuda.stream(streams[r]):  # one stream per rank, as one process per GPU would have
                    chains[r].execute_sharded(chains[r].bind_frame(i, f, ibl, sa, outs[r]))
                streams[r].synchronize()
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        torch.cuda.synchronize()
        for r in range(world):
            assert torch.equal(outs[r][cuts[r]:cuts[r + 1]], want[cuts[r]:cuts[r + 1]]), f"frame {i}: band of rank {r} differs from the unsharded frame"
            assert chains[r].auto_exposure_average() == ref.auto_exposure_average()
        # the history the next frame reprojects: equal to the unsharded chain's on the band and its halo
        for name in ("taa_history", "ssr_history_radiance", "ssr_history_variance", "ssao_history_ao", "ssao_history_len"):
            full = ref_plane(ref, name)
            for r in range(world):
                halo = 8
                lo, hi = max(cuts[r] - halo, 0), min(cuts[r + 1] + halo, h)
                assert torch.equal(ref_plane(chains[r], name)[lo:hi], full[lo:hi]), f"frame {i}: {name} of rank {r}"
        # round 6: the last level of SSAO's depth pyramid, reduced by its owners and all-gathered, is whole and equal to the unsharded chain's on every rank
        if mode != "half resolution" and w % 16 == 0 and h % 16 == 0:
            l4 = ref.effect("ssao").get_intermediate("prefiltered_depth4")
            for r in range(world):
                assert torch.equal(chains[r].effect("ssao").get_intermediate("prefiltered_depth4"), l4), f"frame {i}: level 4 of the prefiltered depth on rank {r}"
    # mifx_comm_get_stats: what left one endpoint arrived at another; the groups of the timed frame came back with a duration each
    after = [c.stats() for c in comms]
    sent = sum(a["bytes_sent"] - b["bytes_sent"] for a, b in zip(after, before))
    received = sum(a["bytes_received"] - b["bytes_received"] for a, b in zip(after, before))
    assert sent == received > 0, (sent, received)
    groups = [a["groups"] - b["groups"] for a, b in zip(after, before)]
    assert len(set(groups)) == 1 and groups[0] >= 3 * len(frames), groups  # every rank issues the same groups: SSAO halos, Bloom gather, TAA + SSR halos (+ the level-4 gather, + the luminance rows)
    per_frame = groups[0] // len(frames)
    assert all(a["timed_groups"] == per_frame and a["exchange_ms_total"] >= a["exchange_ms_max"] > 0.0 for a in after), [(a["timed_groups"], a["exchange_ms_total"]) for a in after]
    assert all(c.stats()["timed_groups"] == 0 for c in comms)  # (read once, then forgotten)
    for r in range(world):
        chains[r].set_sharding(None)
        comms[r].close()
        chains[r].close()
    ref.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["", "depth of field + auto exposure", "three lanes", "three lanes + depth of field + auto exposure"])
def test_sharded_two_lanes_with_frames_queued_back_to_back(mifx_lib, mode):
    """mifx_chain_set_overlap 2 under mifx_chain_execute_sharded: phases 0 - 2 of a frame on the chain's side stream, phase 3 on the context's stream beside the next frame's
    first phases.  Every rank queues all its frames without a synchronisation in between (its own target per frame), at a size whose kernels outlast the host's launches:
    bands and histories equal the unsharded chain's bit for bit."""
    from diligentfx_amd import api

    world, (w, h) = 3, (960, 1080)
    cuts = [0, 330, 700, h]
    ref, ibl, sa, scene, (sobol, tile), torch = _setup(w, h)
    frames = _frames(ref, scene, 7, w, h)
    if "depth of field" in mode:
        for fr in frames:
            fr["camera"].fFocusDistance, fr["camera"].fFStop, fr["camera"].fFocalLength = 12.0, 1.2, 135.0
    max_motion = int(max(float(f["motion"][..., 1].abs().max()) for f in frames) * 0.5 * h) + 2
    chains = [api.Chain(0, sobol, tile) for _ in range(world)]
    for c in chains + [ref]:
        if "auto exposure" in mode:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
        if "depth of field" in mode:
            from diligentfx_amd import binding as B

            da = B.DOFAttribs.default()
            da.MaxCircleOfConfusion = 0.02
            c.set_depth_of_field(da, 3)
    comms = api.Comm.local_group(chains[0].postfx, world)
    for r in range(world):
        chains[r].set_sharding(comms[r], cuts, max_motion)
        chains[r].set_overlap(3 if "three lanes" in mode else 2)
    want = [torch.zeros(h, w, 4, device=ref.device) for _ in frames]
    outs = [[torch.zeros(h, w, 4, device=ref.device) for _ in frames] for _ in range(world)]
    for i, f in enumerate(frames):
        ref.execute(ref.bind_frame(i, f, ibl, sa, want[i]))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(device=ref.device) for _ in range(world)]
    errors = []

    def run(r):
        try:
            with torch.cuda.stream(streams[r]):
                for i, f in enumerate(frames):
                    chains[r].execute_sharded(chains[r].bind_frame(i, f, ibl, sa, outs[r][i]))
            streams[r].synchronize()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(300)
    assert not errors, errors
    torch.cuda.synchronize()
    for r in range(world):
        for i in range(len(frames)):
            assert torch.equal(outs[r][i][cuts[r]:cuts[r + 1]], want[i][cuts[r]:cuts[r + 1]]), f"frame {i}: band of rank {r} differs from the unsharded frame"
    for name in ("taa_history", "ssr_history_radiance", "ssr_history_variance", "ssao_history_ao", "ssao_history_len"):
        full = ref_plane(ref, name)
        for r in range(world):
            lo, hi = max(cuts[r] - 8, 0), min(cuts[r + 1] + 8, h)
            assert torch.equal(ref_plane(chains[r], name)[lo:hi], full[lo:hi]), f"{name} of rank {r}"
    for r in range(world):
        chains[r].set_sharding(None)
        comms[r].close()
        chains[r].close()
    ref.close()


@pytest.mark.gpu
def test_comm_self_test_in_process_group(mifx_lib):
    """mifx_comm_self_test over the in-process group (one host thread and one stream per rank): every rank verifies the slab of every peer; a size two ranks disagree on
    is refused with the sizes in the message."""
    import torch

    from diligentfx_amd import api, binding as B
    from util import blue_noise_tables

    world = 4
    sobol, tile = blue_noise_tables()
    ctxs = [api.PostFXContext(0, sobol, tile) for _ in range(world)]
    comms = api.Comm.local_group(ctxs[0], world)
    streams = [torch.cuda.Stream() for _ in range(world)]
    for sizes, expect_ok in (([1 << 16] * world, True), ([4096, 8192, 8192, 8192], False)):
        results = [None] * world

        def run(r):
            try:
                with torch.cuda.stream(streams[r]):
                    ctxs[r].sync_stream()
                    comms[r].self_test(ctxs[r], sizes[r], timeout_ms=20000)
                results[r] = "ok"
            except B.MifxError as e:
                results[r] = repr(e)

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(200)
        if expect_ok:
            assert results == ["ok"] * world, results
        else:
            assert any("expects" in str(x) and "bytes" in str(x) for x in results), results
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()


def ref_plane(chain, name):
    # mifx_chain_get_shard_plane also serves an unsharded chain (the planes exist either way)
    return chain.shard_plane(name)
