"""Frame runner used by bench.py: builds the synthetic inputs of one GPU's share of the frame, steps the chain, times the passes.

world == 1: the whole frame on one GPU.
world  > 1: every rank renders its own view of --width x --height pixels (weak scaling); see sharding_note()."""
import os
import time

import numpy as np
import torch

from . import api, binding as B, synth

ALGO_BPP = {"pbr_shade": 84.0, "prep": 28.0, "ssr": 327.7, "ssao": 148.0, "composite": 116.0, "taa": 64.0, "dof": 0.0, "bloom": 74.7, "tonemap": 32.0}


def cost_weighted_cuts(depth, world, min_rows, sky_cost=0.34, roughness=None, roughness_threshold=0.2, reflective_cost=None, ghost_rows=60):
    """Row boundaries of `world` bands of about equal cost.  A texel costs `sky_cost` where it is background and 1 where it is geometry; with `roughness` (the plane
    SSR takes its roughness from) a geometry texel that SSR traces -- roughness <= threshold, SSR_Common.fxh:57-60 -- costs `reflective_cost` instead: three classes
    (sky runs the shade, composite, TAA and Bloom; geometry adds SSAO; a reflection sample adds the four SSR passes).  A band also pays for `ghost_rows` rows of its
    neighbours' content on either side (the row windows of the passes).  Weights fitted to tools/shard_cost.py timings at 7680x4320 / 8 ranks (48 bands, rms error
    0.03 ms of 1.3): band time = 0.22 ms + rows x (0.86 us sky, 1.54 us geometry, 2.24 us reflection sample).  Computed from planes every rank holds in full, in
    double precision on the host, so all ranks arrive at the same cuts without communicating.  Bands are at least `min_rows` high."""
    import numpy as np

    h = depth.shape[0]
    is_geom = depth < 1.0 - 1e-6
    geom = is_geom.float().mean(dim=1)            # fraction of geometry texels per row
    w = sky_cost + (1.0 - sky_cost) * geom
    three = roughness is not None and reflective_cost is not None
    if three:
        refl = (is_geom & (roughness <= roughness_threshold)).float().mean(dim=1)
        w = w + (reflective_cost - 1.0) * refl
    w = w.double().cpu()
    if not three:  # the two-class model of round 1: equal sums of the row weights
        cum = torch.cumsum(w, 0)
        total = float(cum[-1])
        cuts = [0]
        for r in range(1, world):
            target = total * r / world
            y = int(torch.searchsorted(cum, torch.tensor(target, dtype=cum.dtype)))
            y = max(y, cuts[-1] + min_rows)
            y = min(y, h - (world - r) * min_rows)
            cuts.append(y)
        cuts.append(h)
        return tuple(cuts)
    # three classes + ghost rows: the smallest T such that `world` bands of cost <= T (own rows + ghost rows on both sides) cover the frame
    cum = np.concatenate([[0.0], np.cumsum(w.numpy())])

    def cost(a, b):
        return cum[min(b + ghost_rows, h)] - cum[max(a - ghost_rows, 0)]

    def cuts_for(t):
        cuts = [0]
        for r in range(1, world):
            a = cuts[-1]
            lo, hi = min(a + min_rows, h - (world - r) * min_rows), h - (world - r) * min_rows
            while lo < hi:                      # the furthest end whose band still costs <= t
                mid = (lo + hi + 1) // 2
                if cost(a, mid) <= t:
                    lo = mid
                else:
                    hi = mid - 1
            cuts.append(lo)
        cuts.append(h)
        return cuts

    lo, hi = 0.0, float(cum[-1])
    for _ in range(60):
        t = 0.5 * (lo + hi)
        c = cuts_for(t)
        if cost(c[-2], h) <= t:
            hi = t
        else:
            lo = t
    return tuple(cuts_for(hi))


# relative to a geometry texel that SSR does not trace.  Round 4 refit with the round's kernels (the hit fetch of the sharded SSR, the parity-first build): 24 bands of three
# cut sets at 7680x4320 / 8 ranks, band time = 0.17 ms + rows x (0.80 us sky, 1.44 us geometry, 2.31 us reflection sample), rms error 0.027 ms
# (profiles/r04_shard_cost_fit.txt; round 2's constants, 0.561 / 1.458, left 0.063 ms): the slowest band 1.369 -> 1.341 ms.
SKY_COST, REFLECTIVE_COST = 0.557, 1.607


def refine_cuts(cuts, times_ms, height, min_rows, fixed_ms=0.17, damping=1.0):
    """Band heights fed back from MEASURED band times (round 5): what the three-class cost model above cannot see -- ray lengths, hits outside the band, how much of a band's
    ghost rows is sky -- is +-0.05 ms per band.  Every band's time above the fixed per-rank part is spread evenly over its rows (a piecewise-constant cost per row), and the
    new cuts give every band the same share of the total.  A pure function of (cuts, times): every rank computes the same cuts from the all-gathered times.  `damping` < 1
    moves the cuts only part of the way (the model ignores that a band's ghost rows move with its cuts)."""
    world = len(times_ms)
    assert len(cuts) == world + 1
    dens = [max(float(times_ms[i]) - fixed_ms, 1e-3) / float(cuts[i + 1] - cuts[i]) for i in range(world)]
    total = sum(dens[i] * (cuts[i + 1] - cuts[i]) for i in range(world))
    new, band, acc = [0], 0, 0.0
    for r in range(1, world):
        target = total * r / world
        while band < world - 1 and acc + dens[band] * (cuts[band + 1] - cuts[band]) < target:
            acc += dens[band] * (cuts[band + 1] - cuts[band])
            band += 1
        y = cuts[band] + (target - acc) / dens[band]
        y = cuts[r] + damping * (y - cuts[r])
        y = int(round(y))
        y = max(y, new[-1] + min_rows)
        y = min(y, height - (world - r) * min_rows)
        new.append(y)
    new.append(height)
    return tuple(new)


def band_cuts(frame, ssr_attribs, world, min_rows, sky_cost=None, reflective_cost="default"):
    """cost_weighted_cuts for a resident frame: depth + the plane / channel / threshold SSR takes its reflection samples from."""
    kw = {}
    rc = REFLECTIVE_COST if reflective_cost == "default" else reflective_cost
    if sky_cost is not None:
        kw["sky_cost"] = sky_cost
    elif rc is not None:
        kw["sky_cost"] = SKY_COST
    if rc is not None:
        kw.update(roughness=frame["material"][..., int(ssr_attribs.RoughnessChannel)].float(), roughness_threshold=float(ssr_attribs.RoughnessThreshold), reflective_cost=rc)
    return cost_weighted_cuts(frame["depth"], world, min_rows, **kw)


class TiledChain:
    def __init__(self, device_index, sobol, tile, rank, world, width, height, shard_rows=False, weighted_bands=True, verify=False, comm_backend="rccl", cuts=None, fallback_backend="nccl"):
        """shard_rows=False: every rank renders its own width x height view (weak scaling, no collective).
        shard_rows=True: the ranks share ONE width x height frame by row bands.  comm_backend "rccl": the exchanges inside the library
        (mifx_chain_execute_sharded: grouped ncclSend / ncclRecv); "torch": driven from Python over torch.distributed (sharded.py; the path the
        gloo tests cover), also the fallback when the library's communicator cannot be created."""
        self.rank, self.world, self.w, self.h = rank, world, width, height
        self.shard_rows = bool(shard_rows) and world > 1
        self.comm_backend, self.mifx_comm, self.comm_note = comm_backend, None, None
        self.weighted_bands, self.cuts, self.fixed_cuts, self.min_rows = weighted_bands, None, cuts, 1
        self.fallback_backend = fallback_backend  # the group the rows get when the library's communicator is unavailable and the side channel is not nccl (None: the side channel)
        self.data_group, self.self_test_note = None, None  # torch.distributed group of the fallback exchanges (None = the default group); what the start-up self test said
        # verify: every rank also runs the unsharded chain on the same frames and compares its band of the output bit for bit
        self.ref_chain = api.Chain(device_index, sobol, tile) if (verify and self.shard_rows) else None
        self.options = []  # option setters (callables taking a chain) that change the image: applied to the verification chain as well (apply_option)
        self.ref_out, self.ref_bound, self.mismatches = None, None, 0
        self.sharded, self.comm = None, None
        self.tables = (sobol, tile)
        self.chain = api.Chain(device_index, sobol, tile)
        self.chain.postfx.set_static_ibl(True)  # the IBL maps of a run are precomputed once: the shade keeps its apron copy of them (mifx_postfx_set_static_ibl)
        self.dev = self.chain.device
        self.scene = synth.Scene()
        self.frames = []
        self.out = None

    def sharding_note(self):
        if self.world == 1:
            return "single GPU, whole frame"
        if self.shard_rows:
            return (f"{self.world} GPUs share one {self.w}x{self.h} frame by row bands ({'cost-weighted, cuts ' + str(list(self.cuts)) if self.cuts else str(self.h // self.world) + ' rows each'}): redundant ghost-row compute, RCCL "
                    f"no radiance exchange (SSR hit colours shaded on demand), gather of Bloom level 1, halo exchange of 5 history planes (max motion {self.max_motion} rows)"
                    + ("; inside the library also Bloom's level-0 rows beside the band edges and the all-gather of SSAO's last depth-pyramid level (round 6)" if self.mifx_comm is not None else "")
                    + f"; {self.comm_note or 'exchanges over torch.distributed (sharded.py)'}")
        return f"{self.world} GPUs, one {self.w}x{self.h} view per GPU (independent frames, no data-path collective)"

    # ------------------------------------------------------------------ inputs
    def build_inputs(self, n_frames=24, first_frame=16):
        """Pre-renders a real camera orbit: `n_frames` CONSECUTIVE camera positions (steady state: frames >= 16), resident in HBM
        (68 B/px per frame: 24 frames of 3840x2160 = 13.5 GB).  step() walks the orbit forwards and, at its end, back again, so that every
        frame's previous-frame depth, previous camera and motion vectors belong to the frame that actually ran before it -- the history the
        temporal passes (A5 / A7 / R6 / T1) reproject is the one the previous step wrote.  Per position the G-buffer is rendered once; the
        motion vectors exist for both directions of travel (they depend on the pair of cameras)."""
        w, h, dev = self.w, self.h, self.dev
        # each rank looks at the scene from its own orbit phase (rank-dependent first frame) so that the ranks do not render identical pixels
        base = first_frame + (0 if self.shard_rows else 40 * self.rank)  # a shared frame: every rank holds the same full-frame inputs
        n_frames = max(int(n_frames), 2)
        cams = [synth.make_camera(base + i, w, h) for i in range(n_frames)]
        self.frames = []
        for i in range(n_frames):
            prev_c, next_c = cams[max(i - 1, 0)], cams[min(i + 1, n_frames - 1)]
            g = synth.render_gbuffer(self.scene, cams[i], prev_c, w, h, dev, also_relative_to=next_c)
            g["motion_fwd"], g["motion_bwd"] = g.pop("motion"), g.pop("motion_alt")
            for k in ("base_color", "normal", "material"):  # 4-channel inputs in the storage of the loaded library (float16 with MIFX_STORAGE=h4)
                g[k] = B.to_storage(g[k])
            g["camera"] = cams[i]
            self.frames.append(g)
        # the first position is only ever entered travelling backwards (or as the very first frame, where every temporal pass resets anyway)
        env = synth.make_sky_cube(256, dev)
        self.ibl = api.precompute_ibl(self.chain.postfx, env)  # reference defaults: LUT 512^2/512, irradiance 64^2/8192, prefiltered 256^2 x 9 mips/256
        self.chain.postfx.set_static_ibl(True)  # (new maps, possibly at the addresses of the old ones: the shade's apron copy is re-made at the next call)
        self.shade = synth.make_lights()
        self.shade.PrefilteredCubeLastMip = float(len(self.ibl.pre) - 1)
        self.out = torch.empty(h, w, 4, device=dev, dtype=B.storage_dtype())
        self.bound = {}
        self.t = 0  # steps taken so far: FrameDesc.Index = 1000 + t, consecutive over warm-up, timed region and the per-stage sweep
        if self.shard_rows:
            from . import sharded

            # bound on the reprojection reach in rows, from the motion vectors of the resident frames (+ 2 rows of slack)
            self.max_motion = int(max(max(float(f["motion_fwd"][..., 1].abs().max()), float(f["motion_bwd"][..., 1].abs().max())) for f in self.frames) * 0.5 * h) + 2
            self.min_rows = min(192, h // self.world)  # no band thinner than this: the cost-weighted cuts and their refinement from measured times share the constraint
            if self.fixed_cuts is not None:  # (a recorded run replayed: bench.py --cuts)
                c = tuple(int(v) for v in self.fixed_cuts)
                if len(c) != self.world + 1 or c[0] != 0 or c[-1] != h or any(a >= b for a, b in zip(c, c[1:])):
                    raise ValueError(f"cuts {c} do not partition {h} rows into {self.world} bands")
                self.cuts = c
            elif self.weighted_bands:
                self.cuts = band_cuts(self.frames[0], self.chain.ssr_attribs, self.world, min_rows=self.min_rows)
            elif h % self.world != 0:
                raise ValueError("equal bands need a height divisible by the number of ranks")
            if self.ref_chain is not None:
                self.ref_out = torch.empty(h, w, 4, device=dev, dtype=B.storage_dtype())
                self.ref_bound = {}
            if self.cuts is None:
                self.cuts = tuple(h * r // self.world for r in range(self.world + 1))
            if self.comm_backend == "rccl":
                self._create_mifx_comm()
            if self.mifx_comm is not None:
                self.chain.set_sharding(self.mifx_comm, list(self.cuts), self.max_motion)
                # the library's own exchanges: the sharded frame as three lanes (phase 3 -- Bloom's coarse levels and the final pass -- beside the next frame's shade and
                # SSAO; prep + SSAO beside the shade and SSR: -2.6 % of the slowest band at 8K / 8 ranks against two lanes, profiles/r05_shard_cost_8k_v5_three_lanes_ab.txt);
                # the inputs of every frame are resident before the first step, which is that mode's contract
                self.chain.set_overlap(int(os.environ.get("MIFX_SHARD_OVERLAP", "3")))
            else:
                self.sharded = sharded.ShardedChain(self.chain, h, self.rank, self.world, self.max_motion, self.cuts)
                self.comm = sharded.TorchDistComm(self.rank, self.world, group=self.data_group, cuts=self.cuts)
        torch.cuda.synchronize(dev)

    def time_own_band(self, frames=6, warm=None):
        """Device time of this rank's band per frame with the exchanges left out (the phases of mifx_chain_execute_phase back to back: stale ghost rows do not change the
        work) -- the quantity tools/shard_cost.py reports and refine_cuts() balances.  Leaves the histories in a state only a reset repairs."""
        lib = self.chain.lib
        y0, y1 = self.cuts[self.rank], self.cuts[self.rank + 1]
        self.chain.set_row_band(y0, y1, self.max_motion)
        warm = 2 * len(self.frames) if warm is None else warm
        bound = {}
        base = getattr(self, "_band_t", 0)  # consecutive frame indices over repeated timings: an index gap would reset every history and time another path

        def band_step(i):
            k, kp = self.orbit_position(base + i)
            b = bound.get((k, kp))
            if b is None:
                b = bound[(k, kp)] = self.chain.bind_frame(3000 + base + i, self._frame_view(k, kp), self.ibl, self.shade, self.out)
            b[0].frame.Index = 3000 + base + i
            self.chain.execute_band(b)  # (the phases and -- with mifx_chain_set_overlap >= 2 -- the lanes of execute_sharded)

        for i in range(warm):
            band_step(i)
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(self.dev)
        a.record()
        for i in range(frames):
            band_step(warm + i)
        z.record()
        torch.cuda.synchronize(self.dev)
        self._band_t = base + warm + frames
        del lib
        return a.elapsed_time(z) / frames

    def calibrate_cuts(self, rounds=2, frames=6, repeats=3):
        """Band heights from measured band times: every rank times its own band (time_own_band; the median of `repeats` timings of `frames` frames, the first after a walk of
        the orbit, the others after two more frames), the times are all-gathered, refine_cuts() moves the cuts under the constraint the initial cuts had (self.min_rows), the
        sharding is set up again with them and every history is reset -- `rounds` times, before the warm-up of a run.  Returns the list of (cuts, times) per round for the bench
        line; `--cuts` of bench.py replays a line's final_cuts."""
        import torch.distributed as dist

        from . import sharded

        trail = []
        for _ in range(rounds):
            if self.mifx_comm is not None:
                self.chain.set_sharding(None)
            own = sorted([self.time_own_band(frames)] + [self.time_own_band(frames, warm=2) for _ in range(max(repeats, 1) - 1)])
            t = torch.tensor([own[len(own) // 2]], dtype=torch.float64, device=self.dev if dist.get_backend() == "nccl" else "cpu")
            times = [torch.zeros_like(t) for _ in range(self.world)]
            dist.all_gather(times, t)
            times = [float(x.item()) for x in times]
            trail.append({"cuts": list(self.cuts), "band_ms": [round(x, 4) for x in times]})
            self.cuts = refine_cuts(self.cuts, times, self.h, self.min_rows)
            self.chain.set_row_band(0, 0, 0)
            if self.mifx_comm is not None:
                self.chain.set_sharding(self.mifx_comm, list(self.cuts), self.max_motion)
            else:
                self.sharded = sharded.ShardedChain(self.chain, self.h, self.rank, self.world, self.max_motion, self.cuts)
                self.comm = sharded.TorchDistComm(self.rank, self.world, group=self.data_group, cuts=self.cuts)
            self.chain.reset_history()
            self.bound = {}
        return trail

    def _create_mifx_comm(self):
        """The library's RCCL communicator: rank 0 draws the ncclUniqueId, torch.distributed carries it to the others (any side channel would do).
        Before a frame depends on it every rank exchanges a known slab with every peer through it and verifies what arrives (mifx_comm_self_test: the first contact
        with RCCL at N > 1 is this, not the timed region).  All ranks use the communicator or none does: a failed creation or self test anywhere sends everyone to the
        torch.distributed path, and the bench line says why (self.comm_note carries RCCL's message)."""
        import torch.distributed as dist

        ok, note = 1, "exchanges inside libmifx (mifx_chain_execute_sharded: grouped ncclSend / ncclRecv over xGMI)"
        uid = [None]
        try:
            if self.rank == 0:
                uid[0] = api.Comm.unique_id()
        except B.MifxError as e:
            ok, note = 0, f"mifx_comm_get_unique_id failed ({e}); exchanges over torch.distributed"
        dist.broadcast_object_list(uid, src=0)
        if uid[0] is None:
            ok = 0
        comm = None
        if ok:
            try:
                comm = api.Comm.create(self.chain.postfx, uid[0], self.rank, self.world)
            except B.MifxError as e:
                ok, note = 0, f"mifx_comm_create failed ({e}); exchanges over torch.distributed"
        # (every rank that has a communicator takes part in the self test -- it is collective; a rank without one lets the others run into the time-out)
        if comm is not None:
            try:
                comm.self_test(self.chain.postfx, 1 << 20, timeout_ms=int(os.environ.get("MIFX_COMM_SELF_TEST_TIMEOUT_MS", "60000")))
                note += "; start-up self test passed (1 MiB to and from every peer, verified)"
                self.self_test_note = "passed: 1 MiB to and from every peer through grouped ncclSend / ncclRecv, verified word by word"
            except B.MifxError as e:
                ok, note = 0, f"mifx_comm_self_test failed ({e}); exchanges over torch.distributed"
                self.self_test_note = f"failed: {e}"
        flag = torch.tensor([ok], dtype=torch.int32, device=self.dev if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            self.mifx_comm = comm
        else:
            if comm is not None:
                comm.close()
            if ok:
                note = "another rank could not create the library's communicator or failed its self test; exchanges over torch.distributed"
            # the rows then travel over torch.distributed.  When the side channel is gloo (the default beside the library's own communicator) they get an RCCL group of
            # their own -- gloo would stage every slab through the host; if that cannot be created either, the default group carries them and the line says so
            if dist.get_backend() != "nccl" and self.fallback_backend == "nccl":
                try:
                    self.data_group = dist.new_group(backend="nccl")
                    probe = torch.ones(1, device=self.dev)
                    dist.all_reduce(probe, group=self.data_group)
                    torch.cuda.synchronize(self.dev)
                    note += "; rows over a torch.distributed nccl (RCCL) group created for them"
                except Exception as e:  # noqa: BLE001 -- (two ranks on one GPU, no RCCL at all, ...)
                    self.data_group = None
                    note += f"; rows over the {dist.get_backend()} side channel (no nccl group: {str(e).splitlines()[0][:160]})"
        self.comm_note = note

    def apply_option(self, setter):
        """An option that changes the image (feature flags, depth of field, auto exposure): set on this rank's chain and on the unsharded chain it is verified against."""
        self.options.append(setter)
        setter(self.chain)
        if self.ref_chain is not None:
            setter(self.ref_chain)

    def verify_against_unsharded(self, frames=3):
        """After the timed region: both this rank's sharded chain and an unsharded chain start from a history reset and run the next `frames`
        positions of the orbit; this rank's band of every frame must be bit-identical.  Returns the number of frames that differed."""
        if not self.shard_rows:
            return 0
        if self.ref_chain is None:
            self.ref_chain = api.Chain(self.dev.index or 0, *self.tables)
            for setter in self.options:
                setter(self.ref_chain)
            self.ref_out = torch.empty(self.h, self.w, 4, device=self.dev, dtype=B.storage_dtype())
        self.chain.reset_history()
        self.ref_chain.reset_history()
        y0, y1 = self.cuts[self.rank], self.cuts[self.rank + 1]
        bad = 0
        for _ in range(frames):
            t = self.t
            k, kp = self.orbit_position(t)
            self.step()
            rb = self.ref_chain.bind_frame(1000 + t, self._frame_view(k, kp), self.ibl, self.shade, self.ref_out)
            self.ref_chain.execute(rb)
            torch.cuda.synchronize(self.dev)
            bad += int(not torch.equal(self.out[y0:y1], self.ref_out[y0:y1]))
        return bad

    def time_unsharded_same_frame(self, frames=10, warm=6, overlap=5):
        """The WHOLE width x height frame on this one GPU, in the unsharded chain's best mode (three lanes, two frames in flight: mode 5), on the same orbit: ms per frame.  What a sharded
        run's frame time has to be divided into for a speed-up that compares like with like (bench.py: single_gpu_same_frame_ms).  Uses the verification chain."""
        if self.ref_chain is None:
            self.ref_chain = api.Chain(self.dev.index or 0, *self.tables)
            for setter in self.options:
                setter(self.ref_chain)
            self.ref_out = torch.empty(self.h, self.w, 4, device=self.dev, dtype=B.storage_dtype())
        c = self.ref_chain
        c.postfx.set_static_ibl(True)
        c.set_overlap(overlap)
        bound = {}

        def one(i):
            k, kp = self.orbit_position(i)
            b = bound.get((k, kp))
            if b is None:
                b = bound[(k, kp)] = c.bind_frame(5000 + i, self._frame_view(k, kp), self.ibl, self.shade, self.ref_out)
            b[0].frame.Index = 5000 + i
            c.execute(b)

        for i in range(warm):
            one(i)
        a, z = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(self.dev)
        a.record()
        for i in range(frames):
            one(warm + i)
        z.record()
        torch.cuda.synchronize(self.dev)
        c.set_overlap(0)
        c.reset_history()
        return a.elapsed_time(z) / frames

    def verify_overlap_against_one_stream(self, frames=6, fusion_mask=None):
        """The run's own stream mode against the one-stream chain, bit for bit, at the run's own size on the run's own orbit: both chains start from a history reset; the
        frames of the run's chain are queued back to back WITHOUT a host synchronisation (as in the timed region: the lanes of consecutive frames really slide over each
        other), each into a plane of its own; a second chain object in mode 0 then produces the same frames one by one.  Returns the number of frames that differed."""
        ref = api.Chain(self.dev.index or 0, *self.tables)
        ref.postfx.set_static_ibl(True)
        for setter in self.options:
            setter(ref)
        if fusion_mask is not None:
            ref.set_fusion_mask(fusion_mask)
        ref.set_overlap(0)
        self.chain.reset_history()
        ref.reset_history()
        outs = [torch.empty(self.h, self.w, 4, device=self.dev, dtype=B.storage_dtype()) for _ in range(frames)]
        ref_out = torch.empty_like(outs[0])
        t0, keep = self.t, []
        for i in range(frames):
            k, kp = self.orbit_position(t0 + i)
            b = self.chain.bind_frame(1000 + t0 + i, self._frame_view(k, kp), self.ibl, self.shade, outs[i])
            keep.append(b)
            self.chain.execute(b)
        torch.cuda.synchronize(self.dev)
        bad = 0
        for i in range(frames):
            k, kp = self.orbit_position(t0 + i)
            rb = ref.bind_frame(1000 + t0 + i, self._frame_view(k, kp), self.ibl, self.shade, ref_out)
            ref.execute(rb)
            torch.cuda.synchronize(self.dev)
            bad += int(not torch.equal(outs[i], ref_out))
        self.t = t0 + frames
        self.bound = {}
        ref.close()
        return bad

    def comm_report(self):
        """What carried the rows of this run, for the bench line (rank-local; bench.py adds the per-frame byte and time figures from two stats() readings)."""
        if self.mifx_comm is not None:
            st = self.mifx_comm.stats()
            return {"transport": "libmifx: grouped ncclSend / ncclRecv on the library's own RCCL communicator" + (f" (MIFX_RCCL_PATH={os.environ['MIFX_RCCL_PATH']})" if os.environ.get("MIFX_RCCL_PATH") else ""),
                    "self_test": self.self_test_note, "ranks_in_communicator": st["ranks_in_communicator"] if st["ranks_in_communicator"] >= 0 else None, "world": st["world"], "is_rccl": bool(st["is_rccl"])}
        import torch.distributed as dist

        grp = self.data_group
        return {"transport": f"torch.distributed ({dist.get_backend(grp) if grp is not None else dist.get_backend()}): sharded.py", "self_test": self.self_test_note,
                "ranks_in_communicator": dist.get_world_size(grp) if grp is not None else dist.get_world_size(), "world": self.world, "is_rccl": False}

    def orbit_position(self, t):
        """(position, previous position) of step t on the forwards-and-back walk over the resident orbit: 0 1 .. n-1 n-2 .. 1 0 1 .."""
        n = len(self.frames)
        period = 2 * (n - 1)

        def pos(k):
            k %= period
            return k if k < n else period - k

        return pos(t), (pos(t - 1) if t > 0 else pos(t))

    def _frame_view(self, k, kp):
        """The resident G-buffer of position k as the frame that follows position kp."""
        f = dict(self.frames[k])
        f["motion"] = f["motion_fwd"] if kp <= k else f["motion_bwd"]
        f["prev_depth"] = self.frames[kp]["depth"]
        f["prev_camera"] = self.frames[kp]["camera"]
        return f

    def step(self, i=None):
        """One frame of the chain: the next position of the orbit.  Frame indices are consecutive from the first call on (history is kept);
        `i` is accepted for compatibility and ignored."""
        t = self.t
        self.t += 1
        k, kp = self.orbit_position(t)
        b = self.bound.get((k, kp))
        if b is None:  # the descriptors of a resident (position, direction) are built once; only the frame index changes from step to step
            b = self.bound[(k, kp)] = self.chain.bind_frame(1000 + t, self._frame_view(k, kp), self.ibl, self.shade, self.out)
        b[0].frame.Index = 1000 + t
        self.last_frame = b[3]  # the G-buffer dict of the frame being executed (tools)
        if self.mifx_comm is not None:
            self.chain.execute_sharded(b)
        elif self.sharded is not None:
            self.sharded.step(b, self.comm)
            if self.ref_chain is not None and self.ref_bound is not None:
                rb = self.ref_bound.get((k, kp))
                if rb is None:
                    rb = self.ref_bound[(k, kp)] = self.ref_chain.bind_frame(1000 + t, self._frame_view(k, kp), self.ibl, self.shade, self.ref_out)
                rb[0].frame.Index = 1000 + t
                self.ref_chain.execute(rb)
                y0, y1 = self.sharded.band
                self.mismatches += int(not torch.equal(self.out[y0:y1], self.ref_out[y0:y1]))
        else:
            self.chain.execute(b)

    # ------------------------------------------------------------------ per-stage timing (HIP events recorded inside mifx_chain_execute)
    STAGES = ("pbr_shade", "prep", "ssr", "ssao", "composite", "taa", "dof", "bloom", "tonemap")

    def arm_kernel_timing(self, kernel_name, slots):
        """HIP-event bracket around the next `slots` launches of `kernel_name` on the launch stream (mifx_postfx_set_kernel_timing)."""
        import ctypes

        B.check(self.chain.lib.mifx_postfx_set_kernel_timing(self.chain.postfx.handle, kernel_name.encode() if kernel_name else None, ctypes.c_uint32(slots)))

    def kernel_times_ms(self, capacity):
        import ctypes

        buf, n = (ctypes.c_float * capacity)(), ctypes.c_uint32(0)
        B.check(self.chain.lib.mifx_postfx_get_kernel_times(self.chain.postfx.handle, buf, ctypes.c_uint32(capacity), ctypes.byref(n)))
        return [buf[i] for i in range(n.value)]

    def time_passes(self, reps=10):
        """Average per-stage device time of `reps` steady-state frames, measured by the chain itself with HIP events on the launch stream."""
        import ctypes

        lib = self.chain.lib
        B.check(lib.mifx_chain_set_profiling(self.chain.handle, ctypes.c_int32(1)))
        acc = [0.0] * len(self.STAGES)
        buf = (ctypes.c_float * len(self.STAGES))()
        for i in range(reps):
            self.step()
            B.check(lib.mifx_chain_get_stage_times(self.chain.handle, buf))
            for k in range(len(self.STAGES)):
                acc[k] += buf[k]
        B.check(lib.mifx_chain_set_profiling(self.chain.handle, ctypes.c_int32(0)))
        return {n: {"ms": acc[k] / reps, "algo_bytes": ALGO_BPP[n] * self.w * self.h} for k, n in enumerate(self.STAGES)}


class StageRunner(TiledChain):
    """One effect of the hot path on the bench's own orbit (BASELINE configs[1] / [2]) instead of the whole chain:
    mode "ssao": PostFXContext::Execute + ScreenSpaceAmbientOcclusion::Execute on the depth + normal G-buffer (reprojected depth, closest motion, blue noise, then
                 A2 .. A8 with their temporal history) -- 28 + 148 = 176 algorithmic B/px;
    mode "pbr":  the PBR shading entry mifx_pbr_shade_execute (GGX + IBL, radiance + specular-IBL targets) on the G-buffer -- 84 B/px.
    The descriptors of every (position, direction) of the orbit are built once; a step is three or four C-ABI calls."""

    def __init__(self, mode, device_index, sobol, tile, width, height):
        super().__init__(device_index, sobol, tile, 0, 1, width, height)
        assert mode in ("ssao", "pbr")
        self.mode = mode
        self.ctx = self.chain.postfx
        self.ssao = api.ScreenSpaceAmbientOcclusion(self.ctx) if mode == "ssao" else None
        self.ssao_attribs = B.SSAOAttribs.default()

    def build_inputs(self, n_frames=24, first_frame=16):
        super().build_inputs(n_frames, first_frame)
        if self.mode == "pbr":
            self.radiance = torch.empty(self.h, self.w, 4, device=self.dev, dtype=B.storage_dtype())
            self.spec = torch.empty(self.h, self.w, 4, device=self.dev, dtype=B.storage_dtype())

    def _bind(self, k, kp):
        import ctypes

        g = self._frame_view(k, kp)
        keep = [g]
        img = lambda t: keep.append(B.image(t)) or keep[-1]  # noqa: E731
        ptr = ctypes.pointer
        if self.mode == "ssao":
            frame = B.FrameDesc(0, self.w, self.h, self.w, self.h)
            pa = B.PostFXRenderAttribs(ptr(img(g["depth"])), ptr(img(g["prev_depth"])), ptr(img(g["motion"])), ptr(g["camera"]), ptr(g["prev_camera"]))
            sa = B.SSAORenderAttribs(self.ctx.handle, ptr(img(g["depth"])), ptr(img(g["normal"])), ptr(self.ssao_attribs))
            return (frame, pa, sa, keep)
        gb = B.GBuffer(ptr(img(g["base_color"])), ptr(img(g["normal"])), ptr(img(g["material"])), ptr(img(g["depth"])), None, None)
        return (gb, g["camera"], img(self.radiance), img(self.spec), (ctypes.c_float * 4)(0.02, 0.03, 0.05, 0.0), keep)

    def step(self, i=None):
        import ctypes

        t = self.t
        self.t += 1
        k, kp = self.orbit_position(t)
        b = self.bound.get((k, kp))
        if b is None:
            b = self.bound[(k, kp)] = self._bind(k, kp)
        lib, ctx = self.chain.lib, self.ctx
        if self.mode == "ssao":
            frame, pa, sa, _ = b
            frame.Index = 1000 + t
            B.check(lib.mifx_postfx_prepare(ctx.handle, ctypes.byref(frame), 0))
            B.check(lib.mifx_ssao_prepare(self.ssao.handle, ctx.handle, 0))
            B.check(lib.mifx_postfx_execute(ctx.handle, ctypes.byref(pa)))
            B.check(lib.mifx_ssao_execute(self.ssao.handle, ctypes.byref(sa)))
        else:
            gb, cam, o0, o1, bg, _ = b
            B.check(lib.mifx_pbr_shade_execute(ctx.handle, ctypes.byref(gb), ctypes.byref(cam), ctypes.byref(self.shade), ctypes.byref(self.ibl.struct), bg, ctypes.byref(o0),
                                               ctypes.byref(o1)))
