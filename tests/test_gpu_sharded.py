"""Row-band sharding (diligentfx_amd/sharded.py): N ranks emulated in one process on one GPU -- one chain object per rank, the exchanges
done by copying rows between the ranks' planes exactly as the RCCL calls would.  The rows each rank produces must equal the unsharded
chain's output bit for bit, frame after frame (histories included), and the planes that feed the next frame must be exact on band + halo."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

W, H = 640, 768
MAX_MOTION_ROWS = 12
# (world, width, height): equal bands; 600x750 has odd pyramid levels (375 rows at level 1): the per-level kernels and uneven Bloom ownership
# the last two: FEATURE_FLAG_HALF_RESOLUTION of SSAO and SSR inside the row-band phases (round 3), even and odd half sizes
# "ae": auto exposure on -- the low-resolution luminance rows are exchanged after phase 3, phase 4 reduces them and tone-maps
SHAPES = [(2, 640, 768, None, 0), (3, 640, 768, None, 0), (4, 512, 1536, None, 0), (2, 600, 750, None, 0), (3, 640, 768, (0, 330, 520, 768), 0),  # (uneven bands)
          (3, 640, 768, None, 2), (2, 600, 750, (0, 350, 750), 2), (3, 640, 768, None, "ae"), (2, 600, 750, (0, 350, 750), "ae"),
          (4, 640, 768, (0, 340, 372, 410, 768), 0),  # two bands of 32 / 38 rows: thinner than every history halo, ghost rows come from two ranks away
          (3, 640, 768, None, "dof"), (2, 600, 750, (0, 350, 750), "dof"),  # depth of field (temporal smoothing + Karis weights) between TAA and Bloom
          (3, 640, 768, (0, 330, 520, 768), "layers")]  # mifx_chain_set_material_layers: all five layers + two shadow-mapped lights in the shade and in the SSR hit fetch


class LocalComm:
    """The exchanges of sharded.py over the planes of all emulated ranks at once."""

    def __init__(self, sharded):
        self.sh = sharded
        self.n = len(sharded)

    def band(self, r):
        return self.sh[r].band

    def allgather_rows(self, name):
        planes = [s.chain.shard_plane(name) for s in self.sh]
        for r in range(self.n):
            b, e = self.band(r)
            for q in range(self.n):
                if q != r:
                    planes[q][b:e].copy_(planes[r][b:e])

    def gather_owned_rows(self, name, infos):
        planes = [s.chain.shard_plane(name) for s in self.sh]
        owned = [(i.own_begin, i.own_end) for i in infos]
        # the rows owned by the ranks must tile the level exactly
        assert owned[0][0] == 0 and owned[-1][1] == planes[0].shape[0] and all(owned[i][1] == owned[i + 1][0] for i in range(self.n - 1)), owned
        for r, (b, e) in enumerate(owned):
            for q in range(self.n):
                if q != r:
                    planes[q][b:e].copy_(planes[r][b:e])

    def gather_luminance_rows(self, infos):
        planes = [s.chain.shard_plane("ae_low_res") for s in self.sh]
        owned = [(i.ae_begin, i.ae_end) for i in infos]
        assert owned[0][0] == 0 and owned[-1][1] == 64 and all(owned[i][1] == owned[i + 1][0] for i in range(self.n - 1)), owned
        for r, (b, e) in enumerate(owned):
            for q in range(self.n):
                if q != r and e > b:
                    planes[q][b:e].copy_(planes[r][b:e])

    def exchange_halos(self, name, halos):
        planes = [s.chain.shard_plane(name) for s in self.sh]
        H = planes[0].shape[0]
        for q in range(self.n):
            b, e = self.band(q)
            h = halos[q]
            for lo, hi in ((max(b - h, 0), b), (e, min(e + h, H))):  # every ghost row from the rank that owns it (the neighbour, or the rank beyond a thin neighbour)
                for r in range(self.n):
                    rb, re = self.band(r)
                    s0, s1 = max(lo, rb), min(hi, re)
                    if r != q and s1 > s0:
                        planes[q][s0:s1].copy_(planes[r][s0:s1])


def history_mismatch(chain, ref_chain, info, b, e, H=H, W=W):
    """Names of the history planes of `chain` that differ from the unsharded chain's on the rows [b - halo, e + halo)."""
    from diligentfx_amd.sharded import HISTORY_PLANES

    bad = []
    for name, field in HISTORY_PLANES:
        halo = getattr(info, field)
        got, want = chain.shard_plane(name), ref_chain.shard_plane(name)
        lo, hi = max(b - halo, 0), min(e + halo, H)
        cols = W * (4 if name in ("taa_history", "ssr_history_radiance") else 1)  # the row padding up to the pitch is never written
        if not torch.equal(got[lo:hi, :cols], want[lo:hi, :cols]):
            bad.append(name)
    return bad


def run_sharded_frame(sharded, comm, bounds, skip=()):
    from diligentfx_amd.sharded import HISTORY_PLANES

    for s, b in zip(sharded, bounds):
        s.phase(b, 0)
    for s, b in zip(sharded, bounds):
        s.phase(b, 1)
    for s, b in zip(sharded, bounds):
        s.phase(b, 2)
    infos = [s.chain.shard_info(b) for s, b in zip(sharded, bounds)]
    assert all(i.gather_level >= 0 for i in infos)
    if "bloom" not in skip:
        comm.gather_owned_rows("bloom_gather", infos)
    for s, b in zip(sharded, bounds):
        s.phase(b, 3)
    if getattr(sharded[0].chain, "auto_exposure", False):
        if "luminance" not in skip:
            comm.gather_luminance_rows(infos)
        for s, b in zip(sharded, bounds):
            s.phase(b, 4)
    if "history" not in skip:
        for name, field in HISTORY_PLANES:
            common = max(getattr(i, field) for i in infos)  # both sides of an exchange move the same number of rows
            comm.exchange_halos(name, [common] * len(infos))
    return infos


@pytest.mark.parametrize("world,W,H,cuts,half", SHAPES)
def test_sharded_chain_equals_unsharded(mifx_lib, world, W, H, cuts, half):
    import chain_util
    from diligentfx_amd import api, synth
    from diligentfx_amd.sharded import ShardedChain
    from util import blue_noise_tables

    sobol, tile = blue_noise_tables()
    dev = torch.device("cuda", 0)
    scene = synth.Scene()
    ref_chain = api.Chain(0, sobol, tile)
    ibl = api.precompute_ibl(ref_chain.postfx, synth.make_sky_cube(32, dev).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32,
                             lut_samples=32, diffuse_samples=32, specular_samples=16)
    shade = chain_util.shade_attribs(len(ibl.pre) - 1)
    ranks = [api.Chain(0, sobol, tile) for _ in range(world)]
    ae, dof, layers, half = half == "ae", half == "dof", half == "layers", 0 if half in ("ae", "dof", "layers") else half
    if layers:
        shade = chain_util.shadowed_shade_attribs(len(ibl.pre) - 1)
        slices, infos_np = chain_util.make_shadow_inputs()
        shadow_maps = torch.from_numpy(np.stack(slices)).to(dev)
    for c in ranks + [ref_chain]:
        c.set_effect_feature_flags(ssao_feature_flags=half, ssr_feature_flags=half)  # 2 = FEATURE_FLAG_HALF_RESOLUTION of both effects
        if ae:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
        if dof:
            from diligentfx_amd import binding as B

            da = B.DOFAttribs.default()
            da.MaxCircleOfConfusion = 0.02
            c.set_depth_of_field(da, 3)
    sharded = [ShardedChain(c, H, r, world, MAX_MOTION_ROWS, cuts) for r, c in enumerate(ranks)]
    comm = LocalComm(sharded)
    out_ref = torch.zeros(H, W, 4, device=dev)
    outs = [torch.full((H, W, 4), -1.0, device=dev) for _ in range(world)]
    prev = None
    for fi in range(16, 22):
        g = synth.make_frame(scene, fi, W, H, dev)
        if dof:
            g["camera"].fFocusDistance, g["camera"].fFStop, g["camera"].fFocalLength = 12.0, 1.2, 135.0
        if layers:  # new planes every frame, whole on every rank like the G-buffer
            from layers_util import IOR, ROTATION, make_layers

            planes, albedo, charlie = make_layers(g["normal"].cpu().numpy(), seed=fi)
            lp = {k: torch.from_numpy(v).to(dev) for k, v in planes.items()}
            lp["transmission"] = lp["transmission"][..., 0].contiguous()
            lp["sheen_albedo_scaling_lut"], lp["preintegrated_charlie"] = torch.from_numpy(albedo).to(dev), torch.from_numpy(charlie).to(dev)
            for c in ranks + [ref_chain]:
                c.set_material_layers(lp, 31, IOR, ROTATION, shadows=(shadow_maps, infos_np, 3))
        m = g["motion"]
        assert float(m[..., 1].abs().max()) * 0.5 * H < MAX_MOTION_ROWS, "the synthetic motion exceeds the declared reprojection reach"
        ref_chain.execute(ref_chain.bind_frame(fi, g, ibl, shade, out_ref))
        bounds = [c.bind_frame(fi, g, ibl, shade, o) for c, o in zip(ranks, outs)]
        infos = run_sharded_frame(sharded, comm, bounds)
        torch.cuda.synchronize()
        for r, s in enumerate(sharded):
            b, e = s.band
            same = torch.equal(outs[r][b:e], out_ref[b:e])
            if not same:
                d = (outs[r][b:e] != out_ref[b:e]).any(dim=-1).nonzero()
                rows = (d[:, 0] + b).unique()
                pytest.fail(f"frame {fi} rank {r}/{world}: {d.shape[0]} pixels differ, rows {rows[:12].tolist()} (band {b}..{e}, info "
                            f"{[(f, getattr(infos[r], f)) for f, _ in infos[r]._fields_]})")
            if ae:  # one average on every rank, the unsharded chain's to the bit
                assert s.chain.auto_exposure_average() == ref_chain.auto_exposure_average(), (fi, r)
            # the rank really worked on its band only: rows of the output outside the band were never written
            assert bool((outs[r][:b] == -1.0).all()) and bool((outs[r][e:] == -1.0).all())
            # what the next frame will read of the histories (band + halo) is exact
            bad = history_mismatch(s.chain, ref_chain, infos[r], b, e, H, W)
            assert not bad, f"frame {fi} rank {r}/{world}: history planes differ on band + halo: {bad}"
        prev = g
    assert prev is not None and np.isfinite(out_ref.cpu().numpy()).all()
    for c in ranks + [ref_chain]:
        c.close()


@pytest.mark.parametrize("skip", ["bloom", "history", "luminance"])
def test_every_exchange_is_needed(mifx_lib, skip):
    """Control: leaving out any one of the exchanges must change the result (otherwise the equality test above proves nothing)."""
    import chain_util
    from diligentfx_amd import api, synth
    from diligentfx_amd.sharded import ShardedChain
    from util import blue_noise_tables

    world = 2
    sobol, tile = blue_noise_tables()
    dev = torch.device("cuda", 0)
    scene = synth.Scene()
    ref_chain = api.Chain(0, sobol, tile)
    ibl = api.precompute_ibl(ref_chain.postfx, synth.make_sky_cube(32, dev).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32,
                             lut_samples=32, diffuse_samples=32, specular_samples=16)
    shade = chain_util.shade_attribs(len(ibl.pre) - 1)
    ranks = [api.Chain(0, sobol, tile) for _ in range(world)]
    if skip == "luminance":
        for c in ranks + [ref_chain]:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
    sharded = [ShardedChain(c, H, r, world, MAX_MOTION_ROWS) for r, c in enumerate(ranks)]
    comm = LocalComm(sharded)
    out_ref = torch.zeros(H, W, 4, device=dev)
    outs = [torch.full((H, W, 4), -1.0, device=dev) for _ in range(world)]
    differs = False
    for fi in range(16, 20):
        g = synth.make_frame(scene, fi, W, H, dev)
        ref_chain.execute(ref_chain.bind_frame(fi, g, ibl, shade, out_ref))
        infos = run_sharded_frame(sharded, comm, [c.bind_frame(fi, g, ibl, shade, o) for c, o in zip(ranks, outs)], skip=(skip,))
        torch.cuda.synchronize()
        differs = differs or any(not torch.equal(outs[r][s.band[0]:s.band[1]], out_ref[s.band[0]:s.band[1]]) for r, s in enumerate(sharded))
        # (a missing history exchange reaches the band only after ghost / motion frames; the planes the next frame reads show it at once)
        differs = differs or any(history_mismatch(s.chain, ref_chain, infos[r], *s.band) for r, s in enumerate(sharded))
    assert differs, f"dropping the {skip} exchange went unnoticed"
    for c in ranks + [ref_chain]:
        c.close()



def band_against_phases(dev, overlap, ae, make_ibl, W=W, H=H, cuts=(0, 250, 520, H), on_frame=None):
    """(also run by tests/cpu_product/run.py `band` on the CPU build of the host code)"""
    import os

    import chain_util
    from diligentfx_amd import api, synth
    from diligentfx_amd.sharded import HISTORY_PLANES, ShardedChain
    from util import blue_noise_tables

    sobol, tile = blue_noise_tables()
    scene = synth.Scene()
    a, b = api.Chain(0, sobol, tile), api.Chain(0, sobol, tile)
    ibl, n_pre = make_ibl(a)
    shade = chain_util.shade_attribs(n_pre - 1)
    for c in (a, b):
        if ae:
            c.set_auto_exposure(True, elapsed_time_s=0.25)
    # Two whole frames first, on both objects: without the exchanges the ghost rows and the other ranks' rows of the gathered Bloom level are never written, and a fresh
    # device allocation holds anything -- afterwards both objects hold the same values there (both slots of every history plane included).
    scratch = torch.zeros(H, W, 4, device=dev)
    for fi in (14, 15):
        g = synth.make_frame(scene, fi, W, H, dev)
        for c in (a, b):
            if on_frame:
                on_frame(g)
            c.execute(c.bind_frame(fi, g, ibl, shade, scratch))
    sa, sb = ShardedChain(a, H, 1, 3, MAX_MOTION_ROWS, cuts), ShardedChain(b, H, 1, 3, MAX_MOTION_ROWS, cuts)
    b.set_overlap(overlap)
    frames = [synth.make_frame(scene, fi, W, H, dev) for fi in range(16, 21)]
    outs_a = [torch.full((H, W, 4), -1.0, device=dev) for _ in frames]
    outs_b = [torch.full((H, W, 4), -1.0, device=dev) for _ in frames]
    for k, g in enumerate(frames):
        if on_frame:
            on_frame(g)
        bound = a.bind_frame(16 + k, g, ibl, shade, outs_a[k])
        for phase in range(ShardedChain.PHASES):
            sa.phase(bound, phase)
    bounds_b = [b.bind_frame(16 + k, g, ibl, shade, outs_b[k]) for k, g in enumerate(frames)]
    # Round 6: a rank of mifx_chain_execute_sharded prefilters only the rows of Bloom's level 0 it owns and receives the few rows beside its band's edges (csrc/api_comm.cpp,
    # mifx_bloom::halo_level0); the phases driven one by one have no such exchange and produce those rows themselves, so the row windows in front of Bloom differ by 9 rows.
    # The equality below is about the phase machinery and the lanes: with the switch off both sides compute the same windows.  (The exchange itself is held to the
    # unsharded frame by the group tests: test_comm.py, the in-process groups below, tests/cpu_product/run.py local_group.)
    before = os.environ.get("MIFX_SHARD_BLOOM_HALO")
    os.environ["MIFX_SHARD_BLOOM_HALO"] = "0"
    try:
        for g, bound in zip(frames, bounds_b):  # no synchronisation in between: the lanes of consecutive frames overlap
            if on_frame:
                on_frame(g)
            b.execute_band(bound)
    finally:
        if before is None:
            del os.environ["MIFX_SHARD_BLOOM_HALO"]
        else:
            os.environ["MIFX_SHARD_BLOOM_HALO"] = before
    if dev.type == "cuda":
        torch.cuda.synchronize()
    lo, hi = sb.band
    for k in range(len(frames)):
        assert torch.equal(outs_a[k][lo:hi], outs_b[k][lo:hi]), f"frame {k}: execute_band differs from the phases"
        assert bool((outs_b[k][:lo] == -1.0).all()) and bool((outs_b[k][hi:] == -1.0).all())
    for name, _ in HISTORY_PLANES:
        assert torch.equal(a.shard_plane(name), b.shard_plane(name)), name
    if ae:
        assert a.auto_exposure_average() == b.auto_exposure_average()
    for c in (a, b):
        c.close()


@pytest.mark.parametrize("overlap,ae", [(0, False), (2, False), (2, True), (3, False), (3, True)])
def test_execute_band_is_the_phases_without_the_exchanges(mifx_lib, overlap, ae):
    """mifx_chain_execute_band (what tools/shard_cost.py and TiledChain.calibrate_cuts time): one rank's band through the phases -- and with overlap >= 2 the two lanes across
    frames -- of mifx_chain_execute_sharded, exchanges left out.  Against a second chain object on the same band driven phase by phase: the band's rows of every frame and all five
    history planes (whole: both hold the same stale ghost rows) are equal bit for bit, frames queued back to back."""
    from diligentfx_amd import api, synth

    dev = torch.device("cuda", 0)

    def make_ibl(chain):
        ibl = api.precompute_ibl(chain.postfx, synth.make_sky_cube(32, dev).clamp(max=200.0), lut_size=32, irradiance_size=8, prefiltered_size=32,
                                 lut_samples=32, diffuse_samples=32, specular_samples=16)
        return ibl, len(ibl.pre)

    band_against_phases(dev, overlap, ae, make_ibl)


def _fake_rccl(tmp_path):
    import os
    import shutil
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = "/opt/rocm"
    if shutil.which("g++") is None or not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        pytest.skip("g++ or the RCCL headers are missing")
    fake = tmp_path / "librccl_fake.so"
    r = subprocess.run(["g++", "-shared", "-fPIC", "-O1", "-std=c++17", "-w", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), os.path.join(root, "tests", "fake_rccl", "fake_rccl.cpp"),
                        "-o", str(fake), "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lrt", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return root, str(fake)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["selftest", "selftest_absent", "selftest_mismatch"])
def test_comm_self_test_and_its_error_paths(tmp_path, mifx_lib, mode):
    """mifx_comm_self_test over the RCCL branch with three processes (the stand-in transport): all ranks -> every rank verifies the slabs of both peers; a rank that never
    posts -> the others get MIFX_ERR_COMM with the transport's message instead of a hang; a rank that announces another slab size -> its peers are told so.  What
    bench.py --gpus N runs before it trusts the communicator with a frame."""
    import os
    import subprocess
    import sys

    root, fake = _fake_rccl(tmp_path)
    env = dict(os.environ, MIFX_RCCL_PATH=fake, HSA_ENABLE_IPC_MODE_LEGACY="0", MIFX_FAKE_RCCL_TIMEOUT="3")
    idfile, world = str(tmp_path / "unique_id"), 3
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "rccl_branch_worker.py"), str(k), str(world), idfile, "64", "64", "0", mode], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for k in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=180)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for k, (rc, o, e) in enumerate(outs):
        assert rc == 0, (k, rc, o[-3000:], e[-1200:])
        if mode == "selftest":
            assert "self test OK" in o, (k, o, e[-800:])
        elif mode == "selftest_absent":
            assert ("stayed away" in o) if k == world - 1 else ("self test FAILED" in o and "MIFX_ERR_COMM" in o), (k, o, e[-800:])
        else:  # rank 0 sends 4096 bytes where 8192 are expected, and expects 4096 where 8192 arrive: every rank that talks to it is refused
            assert "self test FAILED" in o and "MIFX_ERR_COMM" in o, (k, o, e[-800:])

@pytest.mark.parametrize("world,W,H,lanes", [(2, 640, 768, 2), (2, 640, 768, 3), (4, 640, 768, 3), (8, 512, 1536, 2)])
def test_rccl_branch_with_several_processes(tmp_path, mifx_lib, world, W, H, lanes):
    """The RCCL branch of mifx_chain_execute_sharded (csrc/api_comm.cpp: ncclCommInitRank, grouped ncclSend / ncclRecv) with N > 1 ranks: N processes on this one GPU,
    each joining the communicator through mifx_comm_create and running its band; RCCL itself refuses two ranks on one device, so the library loads the stand-in of
    tests/fake_rccl (hipIpc + a shared-memory mailbox behind the same eight entry points) through MIFX_RCCL_PATH.  Every rank compares its band of every frame and its
    history planes on band + halo with the unsharded chain, bit for bit (tests/rccl_branch_worker.py)."""
    import os
    import shutil
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rocm = "/opt/rocm"
    if shutil.which("g++") is None or not os.path.exists(os.path.join(rocm, "include", "rccl", "rccl.h")):
        pytest.skip("g++ or the RCCL headers are missing")
    fake = tmp_path / "librccl_fake.so"
    r = subprocess.run(["g++", "-shared", "-fPIC", "-O1", "-std=c++17", "-w", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(rocm, "include"), os.path.join(root, "tests", "fake_rccl", "fake_rccl.cpp"),
                        "-o", str(fake), "-L", os.path.join(rocm, "lib"), "-lamdhip64", "-lrt", f"-Wl,-rpath,{os.path.join(rocm, 'lib')}"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    env = dict(os.environ, MIFX_RCCL_PATH=str(fake), HSA_ENABLE_IPC_MODE_LEGACY="0", MIFX_SHARD_OVERLAP=str(lanes))  # (mifx_chain_set_overlap of every rank's chain: two or three lanes)
    idfile = str(tmp_path / "unique_id")
    procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "rccl_branch_worker.py"), str(k), str(world), idfile, str(W), str(H), "3"], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.PIPE, text=True) for k in range(world)]
    outs = []
    try:
        for p in procs:
            o, e = p.communicate(timeout=240)
            outs.append((p.returncode, o, e))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for k, (rc, o, e) in enumerate(outs):
        assert rc == 0 and "bit-identical to the unsharded chain" in o and "is_rccl 1" in o, (k, rc, o[-3000:], e[-1200:])
