"""TEST INFRASTRUCTURE ONLY.  Runs scenarios of the DEVICE tests (tests/test_gpu_host_sequence.py: the effect objects driven through the C ABI, compared with oracle/cpu_chain.py)
on the CPU build of the product's host code (tests/cpu_product/build.py) with the reference's shaders standing in for the kernels (tests/cpu_product/device.py).

    MIFX_LIB_PATH=tests/cpu_product/_build/libmifx_cpu.so python tests/cpu_product/run.py scenarios | dof | chain | random FIRST LAST | chain_random FIRST LAST

(started by tests/test_cpu_product.py in a process of its own: the Python mirror diligentfx_amd/api.py is used as it is, with three of its device plumbing points replaced
here -- the stream handle, the device of the tensors, the view of an effect-owned plane -- so that torch CPU tensors stand for device memory.)"""
import ctypes
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests"), HERE]

import numpy as np  # noqa: E402
import torch  # noqa: E402

assert os.environ.get("MIFX_LIB_PATH", "").endswith("libmifx_cpu.so"), "run with MIFX_LIB_PATH pointing at tests/cpu_product/_build/libmifx_cpu.so"

import device as cpu_device  # noqa: E402
import pyref  # noqa: E402
from diligentfx_amd import api, binding as B  # noqa: E402


def install():
    lib = B.load()
    ref = pyref.ref_lib()
    assert ref is not None, "oracle/_ref is needed"
    dev = cpu_device.Device(ref, "ref_")
    cb_type = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.POINTER(cpu_device.Call))
    cb = cb_type(dev.launch)
    lib.mifx_cpu_set_launch_callback(cb)
    dev._keep = cb
    # ---- the three plumbing points of api.py
    api._stream_ptr = lambda device: ctypes.c_void_p(0)
    torch.cuda.synchronize = lambda *a, **k: None
    init = api.PostFXContext.__init__

    def cpu_init(self, device=0, *a, **k):
        init(self, torch.device("cpu"), *a, **k)

    api.PostFXContext.__init__ = cpu_init
    execute = api.PostFXContext.execute

    def noting_execute(self, curr_depth, prev_depth, motion, curr_camera, prev_camera):
        dev.cam, dev.prev_cam = bytes(curr_camera), bytes(prev_camera)  # (the reference's shaders read the whole CameraAttribs block: device.py)
        return execute(self, curr_depth, prev_depth, motion, curr_camera, prev_camera)

    api.PostFXContext.execute = noting_execute
    chain_init, chain_execute = api.Chain.__init__, api.Chain.execute

    def cpu_chain_init(self, device=0, *a, **k):
        chain_init(self, torch.device("cpu"), *a, **k)

    def noting_chain_execute(self, bound):
        dev.cam, dev.prev_cam = bytes(bound[3]["camera"]), bytes(bound[3]["prev_camera"])
        return chain_execute(self, bound)

    api.Chain.__init__, api.Chain.execute = cpu_chain_init, noting_chain_execute
    dof_execute = api.DepthOfField.execute

    def noting_dof_execute(self, color, depth, attribs):
        dev.dof_attribs = bytes(attribs)  # (the launchers of depth of field carry scalars of the block: device.py)
        return dof_execute(self, color, depth, attribs)

    api.DepthOfField.execute = noting_dof_execute

    def host_view(desc, device):
        c, eb, ts = {B.FORMAT_F32: (1, 4, "<f4"), B.FORMAT_F32X2: (2, 4, "<f4"), B.FORMAT_F32X4: (4, 4, "<f4")}[desc.format]
        pitch_f = desc.pitch_bytes // eb
        flat = np.ctypeslib.as_array((ctypes.c_float * (pitch_f * desc.height)).from_address(desc.data))
        rows = torch.from_numpy(flat).view(desc.height, pitch_f)[:, : desc.width * c]
        return rows.view(desc.height, desc.width) if c == 1 else rows.unflatten(1, (desc.width, c))

    api._view = host_view

    def host_shard_plane(self, name):
        d = B.Image2D()
        B.check(self.lib.mifx_chain_get_shard_plane(self.handle, name.encode(), ctypes.byref(d)))
        pitch_f = d.pitch_bytes // 4
        return torch.from_numpy(np.ctypeslib.as_array((ctypes.c_float * (pitch_f * d.height)).from_address(d.data))).view(d.height, pitch_f)

    api.Chain.shard_plane = host_shard_plane
    return dev


# the kernels of this run ARE the checker's shaders: what the device tests hold to 1e-3 must be equal here, bit for bit -- a difference is a difference of sequencing
def exact(got, want, what="", **_):
    assert got.shape == want.shape, (what, got.shape, want.shape)
    same = np.array_equal(got, want, equal_nan=True)
    assert same, f"{what}: {(got != want).mean():.3e} of the values differ (max {np.nanmax(np.abs(got - want)):.3e})"
    return None, 0.0


def chain_random(lib, seed, exact, steps=14):
    """A random sequence through mifx_chain_execute: resizes, frame-index moves, history resets, TAA flag sets, half-resolution SSR / SSAO, the AO algorithm, and -- what the
    checker has no notion of -- the fusion mask and the stream-overlap mode changing from frame to frame.  Every frame equals the CPU chain (tests/chain_util.py) exactly."""
    import chain_util
    import cpu_chain
    import pyref
    from diligentfx_amd import synth
    from util import blue_noise_tables

    rng = np.random.default_rng(7000 + seed)
    sobol, tile = blue_noise_tables()
    ref = pyref.ref_lib()
    chain = api.Chain(0, sobol, tile)
    ibl_np = chain_util.make_ibl(ref, "ref_")
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]), [torch.from_numpy(m) for m in ibl_np["irradiance"]], [torch.from_numpy(m) for m in ibl_np["prefiltered"]])
    cpu = cpu_chain.CpuChain(ref, "ref_")
    scene = synth.Scene()
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    sizes = [(96, 64), (80, 48), (70, 36), (112, 72)]
    w, h = sizes[0]
    idx, cam_pos = int(rng.integers(0, 40)), int(rng.integers(0, 30))
    for step in range(steps):
        if rng.random() < 0.2:
            w, h = sizes[int(rng.integers(len(sizes)))]
        idx += int(rng.choice([1, 1, 1, 1, 0, 2, 5, -1]))
        idx = max(idx, 0)
        cam_pos += 1
        if rng.random() < 0.15:
            chain.reset_history()
            cpu.reset_history()
        taa_flags = int(rng.choice([2, 2, 2, 0, 5, 7]))
        chain.taa_flags = cpu.taa_flags = taa_flags
        algo = int(rng.choice([0, 0, 1, 2]))
        chain.ssao_attribs.Algorithm = algo
        cpu.algorithm = ("gtao", "hbao", "vbao")[algo]
        chain.set_fusion_mask(int(rng.integers(0, 64)))
        chain.set_overlap(int(rng.integers(0, 6)))  # (4 / 5: the planes between the lanes alternate between two sets)
        f = synth.make_frame(scene, cam_pos, w, h, torch.device("cpu"))
        out = torch.zeros(h, w, 4)
        chain.execute(chain.bind_frame(idx, f, ibl, sa, out))
        g = {k: v.numpy() for k, v in f.items() if isinstance(v, torch.Tensor)}
        sattr = type(chain.ssao_attribs).from_buffer_copy(bytes(chain.ssao_attribs))
        want = run_cpu_frame(cpu, chain_util, g, bytes(f["camera"]), bytes(f["prev_camera"]), idx, ibl_np, sa, sattr)
        exact(np.ascontiguousarray(out.numpy()), want, what=f"chain sequence {seed} step {step} (frame {idx}, {w}x{h}, TAA {taa_flags}, algorithm {algo})")
    chain.close()


SHARD_CASES = [(2, 160, 192, None, 0), (3, 160, 192, (0, 70, 130, 192), 0), (4, 128, 256, None, 0), (2, 150, 186, (0, 90, 186), 0), (3, 160, 192, None, 2),
               (4, 160, 192, (0, 85, 93, 103, 192), 0),  # two bands of 8 and 10 rows: thinner than every history halo, ghost rows come from two ranks away
               (3, 160, 192, None, "dof"), (2, 150, 186, (0, 90, 186), "dof"),  # depth of field (temporal + Karis) between TAA and Bloom
               (4, 64, 640, None, "wide ao")]  # a tall frame and an effect radius of 8: taps of every pyramid level, bands + reaches well inside the frame -- the per-level store
#                                                windows of SSAO's depth pyramid (api_ssao.cpp) bind (with an eighth of the reach this case fails)


def sharded_case(lib, case, frames=4, max_motion_rows=12, skip=()):
    """Row-band sharding on the CPU product: N chain objects, each running the phases of mifx_chain_execute_phase on its band, the exchanges done by copying rows between their
    planes (tests/test_gpu_sharded.py LocalComm) -- against one unsharded chain object.  The launch handlers write only the row windows the product hands them, so a pass that
    reads rows no producer computed shows up as a difference: the bands' rows and the history planes on band + halo must be EQUAL.  (The shade handler writes whole frames and the
    hit fetch is a no-op: device.py.)"""
    import chain_util
    import test_gpu_sharded as S
    from diligentfx_amd import synth
    from diligentfx_amd.sharded import ShardedChain
    from util import blue_noise_tables

    world, W, H, cuts, half = SHARD_CASES[case]
    sobol, tile = blue_noise_tables()
    ref = pyref.ref_lib()
    ibl_np = chain_util.make_ibl(ref, "ref_")
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]), [torch.from_numpy(m) for m in ibl_np["irradiance"]], [torch.from_numpy(m) for m in ibl_np["prefiltered"]])
    shade = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    ref_chain = api.Chain(0, sobol, tile)
    ranks = [api.Chain(0, sobol, tile) for _ in range(world)]
    dof, wide, half = half == "dof", half == "wide ao", 0 if half in ("dof", "wide ao") else half
    for c in ranks + [ref_chain]:
        c.set_effect_feature_flags(ssao_feature_flags=half, ssr_feature_flags=half)
        if wide:
            c.ssao_attribs.EffectRadius = 8.0
        if dof:
            da = B.DOFAttribs.default()
            da.MaxCircleOfConfusion = 0.02
            c.set_depth_of_field(da, 3)
            DEVICE.dof_attribs = bytes(da)
    sharded = [ShardedChain(c, H, r, world, max_motion_rows, cuts) for r, c in enumerate(ranks)]
    comm = S.LocalComm(sharded)
    scene = synth.Scene()
    out_ref = torch.zeros(H, W, 4)
    outs = [torch.full((H, W, 4), -1.0) for _ in range(world)]
    for fi in range(16, 16 + frames):
        g = synth.make_frame(scene, fi, W, H, torch.device("cpu"))
        assert float(g["motion"][..., 1].abs().max()) * 0.5 * H < max_motion_rows
        if dof:
            g["camera"].fFocusDistance, g["camera"].fFStop, g["camera"].fFocalLength = 12.0, 1.2, 135.0
        DEVICE.cam, DEVICE.prev_cam = bytes(g["camera"]), bytes(g["prev_camera"])
        ref_chain.execute(ref_chain.bind_frame(fi, g, ibl, shade, out_ref))
        bounds = [c.bind_frame(fi, g, ibl, shade, o) for c, o in zip(ranks, outs)]
        infos = S.run_sharded_frame(sharded, comm, bounds, skip=skip)
        for r, sc in enumerate(sharded):
            b, e = sc.band
            assert torch.equal(outs[r][b:e], out_ref[b:e]), f"case {case} frame {fi} rank {r}/{world}: rows {(outs[r][b:e] != out_ref[b:e]).any(-1).any(-1).nonzero()[:8].flatten().tolist()} of the band differ"
            written = ((outs[r] != -1.0).any(-1).any(-1)).nonzero().flatten()
            assert bool((outs[r][:b] == -1.0).all()) and bool((outs[r][e:] == -1.0).all()), f"case {case} rank {r}: band {b}..{e}, rows written {int(written.min())}..{int(written.max())}"
            bad = S.history_mismatch(sc.chain, ref_chain, infos[r], b, e, H, W)
            assert not bad, f"case {case} frame {fi} rank {r}/{world}: history planes differ on band + halo: {bad}"
    for c in ranks + [ref_chain]:
        c.close()


GROUP_CASES = [(2, 160, 192, None, ""), (3, 160, 192, (0, 60, 130, 192), ""), (4, 128, 256, None, ""), (2, 160, 192, None, "half resolution"),
               (4, 160, 192, (0, 84, 92, 102, 192), "thin bands"), (3, 160, 192, (0, 60, 130, 192), "depth of field"),
               (3, 160, 192, (0, 60, 130, 192), "three lanes")]  # mifx_chain_set_overlap 3: the event requests of the SSAO lane (streams do nothing here: the host logic)


def local_group_case(lib, case, frames=4):
    """mifx_chain_execute_sharded with the in-library communicator of one process (mifx_comm_create_local_group: csrc/api_comm.cpp -- the same code that decides which rows
    go to whom for the RCCL transport, with a copy in place of ncclSend / ncclRecv): one thread per rank, against the unsharded chain object, for equality
    (tests/test_comm.py::test_sharded_execute_in_process_group on the CPU build)."""
    import threading

    import chain_util
    from diligentfx_amd import synth
    from util import blue_noise_tables

    if case >= 100:  # a random configuration (round 6: the level-4 gather with its compute windows; bands that own no row of the last level; frames not divisible by 16)
        import random

        rng = random.Random(case)
        world = rng.randint(2, 6)
        w, h = rng.choice([(160, 192), (128, 256), (96, 320), (176, 208), (144, 240), (150, 200)])
        inner = sorted(rng.sample(range(4, h - 3), world - 1))
        cuts = (0, *inner, h) if rng.random() < 0.8 else None
        mode = rng.choice(["", "", "three lanes", "three lanes", "half resolution"])
    else:
        world, w, h, cuts, mode = GROUP_CASES[case]
    cuts = list(cuts) if cuts else [h * r // world for r in range(world + 1)]
    sobol, tile = blue_noise_tables()
    ibl_np = chain_util.make_ibl(pyref.ref_lib(), "ref_")
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]), [torch.from_numpy(m) for m in ibl_np["irradiance"]], [torch.from_numpy(m) for m in ibl_np["prefiltered"]])
    sa = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    ref = api.Chain(0, sobol, tile)
    chains = [api.Chain(0, sobol, tile) for _ in range(world)]
    scene = synth.Scene()
    fr = [synth.make_frame(scene, 16 + i, w, h, torch.device("cpu")) for i in range(frames)]
    for c in chains + [ref]:
        if mode == "half resolution":
            c.set_effect_feature_flags(ssao_feature_flags=2, ssr_feature_flags=2)
        if mode == "depth of field":
            da = B.DOFAttribs.default()
            da.MaxCircleOfConfusion = 0.02
            c.set_depth_of_field(da, 3)
            DEVICE.dof_attribs = bytes(da)
    if mode == "depth of field":
        for f in fr:
            f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = 12.0, 1.2, 135.0
    max_motion = int(max(float(f["motion"][..., 1].abs().max()) for f in fr) * 0.5 * h) + 2
    comms = api.Comm.local_group(chains[0].postfx, world)
    outs = [torch.zeros(h, w, 4) for _ in range(world)]
    for r in range(world):
        assert comms[r].info() == (r, world, False)
        chains[r].set_sharding(comms[r], cuts, max_motion)
        if mode == "three lanes":
            chains[r].set_overlap(3)
    want = torch.zeros(h, w, 4)
    for i, f in enumerate(fr):
        DEVICE.cam, DEVICE.prev_cam = bytes(f["camera"]), bytes(f["prev_camera"])
        ref.execute(ref.bind_frame(16 + i, f, ibl, sa, want))
        errors = []

        def run(r):
            try:
                chains[r].execute_sharded(chains[r].bind_frame(16 + i, f, ibl, sa, outs[r]))
            except Exception as e:  # noqa: BLE001
                errors.append((r, repr(e)))

        threads = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
        assert not errors and not any(t.is_alive() for t in threads), errors
        for r in range(world):
            assert torch.equal(outs[r][cuts[r]:cuts[r + 1]], want[cuts[r]:cuts[r + 1]]), f"case {case} frame {i}: the band of rank {r} differs from the unsharded frame (world {world}, {w}x{h}, cuts {cuts}, {mode!r})"
        for name in ("taa_history", "ssr_history_radiance", "ssr_history_variance", "ssao_history_ao", "ssao_history_len"):
            full = ref.shard_plane(name)
            for r in range(world):
                lo, hi = max(cuts[r] - 8, 0), min(cuts[r + 1] + 8, h)
                assert torch.equal(chains[r].shard_plane(name)[lo:hi], full[lo:hi]), f"case {case} frame {i}: {name} of rank {r}"
    stats = [comms[r].stats() for r in range(world)]
    infos = [chains[r].shard_info(chains[r].bind_frame(16 + frames - 1, fr[-1], ibl, sa, outs[r])) for r in range(world)]
    for r in range(world):
        chains[r].set_sharding(None)
        comms[r].close()
        chains[r].close()
    ref.close()
    return stats, infos


def order_run(lib, mode, mask, frames=7, band=None, drop_wait=None, dof=False, size=(96, 64)):
    """One chain object, `frames` frames queued without a host synchronisation in between (as bench.py queues them), under tests/cpu_product/order.py: returns the LaneOrder
    with its findings and the index range of the hipStreamWaitEvent calls of the last frame.  band = (y0, y1): mifx_chain_execute_band on that row band instead."""
    import chain_util
    import order as O
    from diligentfx_amd import synth
    from util import blue_noise_tables

    W, H = size
    sobol, tile = blue_noise_tables()
    ibl_np = chain_util.make_ibl(pyref.ref_lib(), "ref_")
    ibl = api.IBLResources(torch.from_numpy(ibl_np["lut"]), [torch.from_numpy(m) for m in ibl_np["irradiance"]], [torch.from_numpy(m) for m in ibl_np["prefiltered"]])
    shade = chain_util.shade_attribs(len(ibl_np["prefiltered"]) - 1)
    scene = synth.Scene()
    fr = [synth.make_frame(scene, 16 + i, W, H, torch.device("cpu")) for i in range(frames)]
    track = O.LaneOrder()
    track.drop_wait = drop_wait
    cb = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong)(track.runtime)
    lib.mifx_cpu_set_runtime_callback(cb)
    cpu_device.TRACK = track
    try:
        chain = api.Chain(0, sobol, tile)
        if dof:
            da = B.DOFAttribs.default()
            da.MaxCircleOfConfusion = 0.02
            chain.set_depth_of_field(da, 3)
            DEVICE.dof_attribs = bytes(da)
            for f in fr:
                f["camera"].fFocusDistance, f["camera"].fFStop, f["camera"].fFocalLength = 12.0, 1.2, 135.0
        chain.set_fusion_mask(mask)
        if band:
            chain.set_row_band(band[0], band[1], 12)
        chain.set_overlap(mode)
        out = torch.zeros(H, W, 4)
        last = (0, 0)
        for i, f in enumerate(fr):
            DEVICE.cam, DEVICE.prev_cam = bytes(f["camera"]), bytes(f["prev_camera"])
            w0 = track.waits
            b = chain.bind_frame(16 + i, f, ibl, shade, out)
            (chain.execute_band if band else chain.execute)(b)
            last = (w0, track.waits)
        chain.set_overlap(0)
        chain.close()
    finally:
        cpu_device.TRACK = None
        lib.mifx_cpu_set_runtime_callback(ctypes.cast(None, ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong)))
    return track, last


def run_cpu_frame(cpu, chain_util, g, cam, prev, frame_index, ibl, sa, ssao_attribs):
    """chain_util.run_frame_inputs with the SSAO attributes of the frame (the algorithm changes from frame to frame here)."""
    from diligentfx_amd import binding as B
    from util import blue_noise_tables

    h, w = g["depth"].shape
    radiance, spec_ibl = np.zeros((h, w, 4), np.float32), np.zeros((h, w, 4), np.float32)
    cpu.call("pbr_shade", [g["base_color"], g["normal"], g["material"], g["depth"], None, None, ibl["lut"], ibl["irradiance"], ibl["prefiltered"]], [radiance, spec_ibl], cam0=cam,
             attribs=bytes(sa), fval=[0.02, 0.03, 0.05, 0.0])
    pf = cpu.postfx(frame_index, g["depth"], g["prev_depth"], g["motion"], cam, prev, blue_noise_tables())
    ssr = cpu.ssr(pf, radiance, g["depth"], g["normal"], g["material"], g["motion"], B.SSRAttribs.default(), None)
    ssao = cpu.ssao(pf, g["depth"], g["normal"], ssao_attribs, None)
    comp = np.zeros((h, w, 4), np.float32)
    cpu.call("composite", [radiance, spec_ibl, ssr, ssao, g["normal"], g["base_color"], g["material"], ibl["lut"]], [comp], cam0=cam, fval=[1.0, 1.0])
    taa = cpu.taa(pf, comp, B.TAAAttribs.default(), None)
    bloom = cpu.bloom(taa, B.BloomAttribs.default(), None)
    final = np.zeros((h, w, 4), np.float32)
    cpu.call("tonemap", [bloom], [final], attribs=bytes(B.ToneMappingAttribs.default(4)), fval=[0.3], ival=[1])
    return final


DEVICE = None


def main():
    global DEVICE
    dev = DEVICE = install()
    import test_gpu_host_sequence as T

    T.assert_close = exact
    T.to_np = lambda t: np.ascontiguousarray(t.detach().cpu().numpy())  # (a CPU tensor's numpy() is a view of the pitched plane: the device tests get a tight copy)
    what = sys.argv[1] if len(sys.argv) > 1 else "scenarios"
    lib = B.load()
    if what == "scenarios":
        names = sys.argv[2:] or list(T.SCENARIOS)
        for n in names:
            T.test_host_objects_follow_the_reference_sequencing(lib, n)
            print(f"cpu product: scenario OK: {n}", flush=True)
    elif what == "chain":
        import test_gpu_chain as C

        C.assert_close = exact
        C.to_np = T.to_np
        for fn in (sys.argv[2:] or ["test_chain_vs_cpu_chain", "test_chain_reversed_depth"]):
            getattr(C, fn)(lib)
            print(f"cpu product: scenario OK: chain {fn}", flush=True)
    elif what == "arguments":
        # the argument checks of round 5's entries, which need a chain object (the CPU build has one without a GPU)
        from util import blue_noise_tables

        sobol, tile = blue_noise_tables()
        chain = api.Chain(0, sobol, tile)

        def refused(fn, *a):
            try:
                fn(*a)
            except B.MifxError as e:
                return "MIFX_ERR_INVALID_ARG" in str(e) or "INVALID" in str(e)
            return False

        for good in ("", "ssao_compute_ao_kernel<ssr_intersection_kernel@1", "a<b@0,c<d@3,", None):
            chain.set_lane_edges(good)
        for bad in ("a<b", "a@1", "<b@1", "a<@1", "a<b@", "a<b@4", "a<b@-1", "nonsense"):
            assert refused(chain.set_lane_edges, bad), bad
        for mode in range(6):
            chain.set_overlap(mode)
        assert refused(chain.set_overlap, 6) and refused(chain.set_overlap, -1)
        chain.set_fusion_mask(api.Chain.FUSE_EVERY_SWITCH)
        chain.set_fusion_mask(api.Chain.FUSE_DEFAULT)
        assert refused(chain.set_fusion_mask, 64)
        chain.close()
        print("cpu product: scenario OK: arguments of the round-5 entries", flush=True)
    elif what == "layers":
        import test_gpu_pbr_layers as L

        L.assert_close = exact
        L.to_np = T.to_np
        import chain_util

        L.test_chain_with_material_layers(lib, chain_util.make_ibl(pyref.ref_lib(), "ref_"))
        print("cpu product: scenario OK: chain with material layers", flush=True)
    elif what == "sharded":
        for case in range(int(sys.argv[2]), int(sys.argv[3])):
            sharded_case(lib, case)
            print(f"cpu product: sharded case OK: {case}", flush=True)
        for skip in ("bloom", "history"):  # the comparison can see a missing row: without one of the two exchanges the bands differ (cf. test_every_exchange_is_needed)
            try:
                sharded_case(lib, 0, skip=(skip,))
            except AssertionError:
                print(f"cpu product: without the {skip} exchange the bands differ, as they must", flush=True)
            else:
                raise SystemExit(f"the banded run did not notice the missing {skip} exchange")
    elif what == "order":
        # the order of the lanes (order.py): no pair of conflicting accesses of two streams without a happens-before edge, in every stream mode of the chain, with the default
        # fusions, all of them and none, with depth of field, and for one rank's band under the sharded frame's lanes; then the control: every wait of a steady-state frame
        # dropped in turn -- the ones whose absence the handlers' planes can show must be found
        cases = [(m, k, None, False) for m in (0, 1, 2, 3, 4, 5) for k in (31, 63, 0)] + [(3, 31, None, True), (4, 31, None, True), (5, 31, None, True)] + [(m, 31, (16, 48), False) for m in (0, 2, 3)]
        for mode, mask, band, dof in cases:
            t, last = order_run(lib, mode, mask, band=band, dof=dof)
            assert t.launches > 50, t.launches
            assert not t.findings, (mode, mask, band, dof, t.describe()[:6])
            print(f"cpu product: order OK: overlap {mode}, fusion mask {mask}{', band ' + str(band) if band else ''}{', depth of field' if dof else ''}: {t.launches} launches, "
                  f"{t.waits} waits ({last[1] - last[0]} in the last frame), no unordered pair", flush=True)
        for mode, band, dof in ((1, None, False), (2, None, False), (3, None, False), (4, None, False), (5, None, False), (3, None, True), (5, None, True), (2, (16, 48), False), (3, (16, 48), False)):
            base, last = order_run(lib, mode, 31, band=band, dof=dof)
            needed, silent = [], []
            for k in range(*last):
                t, _ = order_run(lib, mode, 31, band=band, dof=dof, drop_wait=k)
                (needed if t.findings else silent).append(k - last[0])
            assert needed, f"overlap {mode}: no dropped wait was noticed -- the check sees nothing"
            print(f"cpu product: order control: overlap {mode}{', band' if band else ''}{', depth of field' if dof else ''}: of the {last[1] - last[0]} waits of a steady-state frame, dropping "
                  f"{len(needed)} leaves an unordered pair {needed}; {len(silent)} are implied by others or guard what this configuration does not touch {silent}", flush=True)
    elif what == "order_random":
        # the random chain sequences (sizes, frame indices, resets, flag sets, the fusion mask and the stream mode changing from frame to frame) under order.py: what the
        # library queues on the context's stream between frames -- history fills of a reset, re-allocations of a resize -- against the lanes of the frames around it
        import order as O

        rt_type = ctypes.CFUNCTYPE(None, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_ulonglong)
        for seed in range(int(sys.argv[2]), int(sys.argv[3])):
            track = O.LaneOrder()
            cb = rt_type(track.runtime)
            lib.mifx_cpu_set_runtime_callback(cb)
            cpu_device.TRACK = track
            try:
                chain_random(lib, seed, exact)
            finally:
                cpu_device.TRACK = None
                lib.mifx_cpu_set_runtime_callback(ctypes.cast(None, rt_type))
            assert not track.findings, (seed, track.describe()[:8])
            print(f"cpu product: order OK: chain sequence {seed}: {track.launches} launches, {track.waits} waits, no unordered pair", flush=True)
    elif what == "band":
        # mifx_chain_execute_band against the phases driven one by one (tests/test_gpu_sharded.py band_against_phases), one stream and two lanes
        import chain_util
        import test_gpu_sharded as S

        def make_ibl(_chain):
            ibl_np = chain_util.make_ibl(pyref.ref_lib(), "ref_")
            return api.IBLResources(torch.from_numpy(ibl_np["lut"]), [torch.from_numpy(m) for m in ibl_np["irradiance"]], [torch.from_numpy(m) for m in ibl_np["prefiltered"]]), len(ibl_np["prefiltered"])

        def on_frame(g):
            DEVICE.cam, DEVICE.prev_cam = bytes(g["camera"]), bytes(g["prev_camera"])

        for overlap in (0, 2, 3):
            S.band_against_phases(torch.device("cpu"), overlap, False, make_ibl, W=160, H=192, cuts=(0, 60, 130, 192), on_frame=on_frame)
            print(f"cpu product: execute_band OK: overlap {overlap}", flush=True)
    elif what == "local_group":
        for case in range(int(sys.argv[2]), int(sys.argv[3])):
            local_group_case(lib, case)
            print(f"cpu product: in-library group OK: {case}", flush=True)
    elif what == "bloom_halo":
        # Round 6, Bloom's level 0 with halos (csrc/api_comm.cpp): the exchange really runs -- one more group per frame -- and buys what it is for: shorter history halos,
        # fewer bytes per frame in total; MIFX_SHARD_BLOOM_HALO=0 (read per frame) is round 5's frame.  Both frames equal the unsharded chain (local_group_case asserts that).
        frames = 2
        os.environ["MIFX_SHARD_BLOOM_HALO"] = "0"
        off, info_off = local_group_case(lib, 1, frames=frames)
        os.environ["MIFX_SHARD_BLOOM_HALO"] = "1"
        on, info_on = local_group_case(lib, 1, frames=frames)
        for a, b, ia, ib in zip(off, on, info_off, info_on):
            assert b["groups"] == a["groups"] + frames, (a, b)
            assert b["bytes_sent"] < a["bytes_sent"] and b["bytes_received"] < a["bytes_received"], (a, b)
            assert ib.halo_taa < ia.halo_taa and ib.halo_ssr < ia.halo_ssr and ib.halo_ssao <= ia.halo_ssao, ((ia.halo_taa, ia.halo_ssr, ia.halo_ssao), (ib.halo_taa, ib.halo_ssr, ib.halo_ssao))
        print(f"cpu product: Bloom level-0 halo OK: groups per frame {off[0]['groups'] // frames} -> {on[0]['groups'] // frames}, bytes sent by rank 1 per frame "
              f"{off[1]['bytes_sent'] // frames} -> {on[1]['bytes_sent'] // frames}, TAA halo {info_off[1].halo_taa} -> {info_on[1].halo_taa} rows", flush=True)
    elif what == "sharded_random":
        for seed in range(int(sys.argv[2]), int(sys.argv[3])):
            rng = np.random.default_rng(9000 + seed)
            world = int(rng.integers(2, 6))
            W, H = [(160, 192), (150, 186), (128, 256), (176, 208), (96, 320)][int(rng.integers(5))]
            inner = np.sort(rng.choice(np.arange(6, H - 6), size=world - 1, replace=False))
            cuts = (0, *[int(c) for c in inner], H) if rng.random() < 0.7 or H % world else None  # (equal bands need a height the ranks divide)
            opt = [0, 0, 2, "dof"][int(rng.integers(4))]
            SHARD_CASES.append((world, W, H, cuts, opt))
            sharded_case(lib, len(SHARD_CASES) - 1, frames=int(rng.integers(2, 5)), max_motion_rows=max(12, H // 16))
            print(f"cpu product: sharded sequence OK: {seed} (world {world}, {W}x{H}, cuts {cuts}, option {opt})", flush=True)
    elif what == "chain_random":
        for seed in range(int(sys.argv[2]), int(sys.argv[3])):
            chain_random(lib, seed, exact)
            print(f"cpu product: chain sequence OK: {seed}", flush=True)
    elif what == "dof":
        T.test_depth_of_field_follows_the_reference_sequencing(lib)
        print("cpu product: scenario OK: depth of field", flush=True)
    elif what == "random":
        for seed in range(int(sys.argv[2]), int(sys.argv[3])):
            T.test_random_sequences_through_the_c_abi(lib, seed)
            print(f"cpu product: random sequence OK: {seed}", flush=True)
    print("cpu product: done; launches:", len(dev.log))


if __name__ == "__main__":
    main()
