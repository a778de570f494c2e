"""world_size > 1 on CPU (gloo): the row-band partition and the two exchange primitives of diligentfx_amd/dist.py, plus the bench's
max-over-ranks timing reduction.  The same code runs over RCCL ("nccl") on the GPU box."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diligentfx_amd.dist import RowBands, allgather_rows, exchange_halos, max_over_ranks


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def worker(rank, world, port, height, width, halo, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(1234)
        full = torch.rand(height, width, 4, generator=g)      # the frame a single GPU would hold
        full1 = torch.rand(height, width, generator=g)
        bands = RowBands(height, world, halo)
        b, e = bands.band(rank)
        # every rank only has its own band
        mine = torch.full_like(full, float("nan"))
        mine[b:e] = full[b:e]
        exchange_halos(mine, bands, rank)
        xb, xe = bands.extended(rank)
        ok_halo = torch.equal(mine[xb:xe], full[xb:xe]) and bool(torch.isnan(mine[:xb]).all()) and bool(torch.isnan(mine[xe:]).all())
        # single-channel plane through the all-gather
        mine1 = torch.zeros_like(full1)
        mine1[b:e] = full1[b:e]
        allgather_rows(mine1, bands, rank)
        ok_gather = torch.equal(mine1, full1)
        # mip-level partition
        m = bands.mip(2)
        ok_mip = m.band(rank) == (b >> 2, e >> 2)
        t = max_over_ranks(1.0 + rank, torch.device("cpu"))
        q.put((rank, ok_halo, ok_gather, ok_mip, t))
    finally:
        dist.destroy_process_group()


# (the last two: a halo taller than a band -- ghost rows come from the rank beyond the neighbour as well)
@pytest.mark.parametrize("world,height,halo", [(2, 64, 8), (3, 96, 5), (2, 32, 16), (4, 64, 24), (3, 96, 40)])
def test_row_band_exchange_gloo(world, height, halo):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=worker, args=(r, world, port, height, 24, halo, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_halo, ok_gather, ok_mip, t in results:
        assert ok_halo, f"rank {rank}: ghost rows differ from the single-GPU frame"
        assert ok_gather, f"rank {rank}: all-gathered plane differs"
        assert ok_mip
        assert t == float(world)  # max over ranks of (1 + rank)


def test_row_bands_partition_math():
    b = RowBands(4320, 8, 64)
    assert b.rows == 540 and b.band(0) == (0, 540) and b.band(7) == (3780, 4320)
    assert b.extended(0) == (0, 604) and b.extended(3) == (1620 - 64, 2160 + 64) and b.extended(7) == (3716, 4320)
    assert b.mip(2).rows == 135
    with pytest.raises(ValueError):
        b.mip(3)  # 540 = 2^2 * 135: bands are not aligned to 8 rows (align or widen the halo)
    with pytest.raises(ValueError):
        RowBands(1081, 2)
    assert RowBands(2160, 1).extended(0) == (0, 2160)


# ---------------------------------------------------------------------------------------------------------------------- sharded.py over gloo
class _FakeInfo:
    def __init__(self, **kw):
        self.__dict__.update(kw)


class _FakeChain:
    """Stands in for api.Chain on the CPU: the planes a rank holds after each phase (its own rows valid, the rest poisoned)."""

    def __init__(self, full, rank, world, height, own, cuts=None):
        self.full, self.rank, self.world, self.height, self.own = full, rank, world, height, own
        self.b, self.e = (cuts[rank], cuts[rank + 1]) if cuts else (rank * height // world, (rank + 1) * height // world)
        self.planes = {}
        self.band = None
        self.device = torch.device("cpu")
        self.auto_exposure = True
        # rows of the 64-row luminance plane this rank writes: rank 1 writes none (a band too thin to hold a sample row)
        lum = {2: (0, 64, 64), 3: (0, 16, 16, 64), 4: (0, 16, 16, 40, 64)}[world]
        self.lum = (lum[rank], lum[rank + 1])

    def set_row_band(self, b, e, m):
        self.band = (b, e, m)

    def execute_phase(self, bound, k):
        nan = float("nan")
        if k in (0, 1):
            pass
        elif k == 2:
            p = torch.full_like(self.full["bloom_gather"], 123.0)  # stale rows of other ranks: must be cleared, not summed
            ob, oe = self.own
            p[ob:oe] = self.full["bloom_gather"][ob:oe]
            self.planes["bloom_gather"] = p
        elif k == 3:
            p = torch.full_like(self.full["ae_low_res"], 7.0)  # (last frame's values on the rows of the other ranks)
            p[self.lum[0]:self.lum[1]] = self.full["ae_low_res"][self.lum[0]:self.lum[1]]
            self.planes["ae_low_res"] = p
        else:
            self.seen_luminance = self.planes["ae_low_res"].clone()  # phase 4 reduces the plane: every row must have arrived by now
            for name in ("taa_history", "ssr_history_radiance", "ssr_history_variance", "ssao_history_ao", "ssao_history_len"):
                p = torch.full_like(self.full[name], nan)
                p[self.b:self.e] = self.full[name][self.b:self.e]
                self.planes[name] = p

    def shard_plane(self, name):
        return self.planes[name]

    def shard_info(self, bound):
        # (every rank reports a different need, as the real chain does: the driver has to agree on the maximum)
        return _FakeInfo(gather_level=2, own_begin=self.own[0], own_end=self.own[1], ae_begin=self.lum[0], ae_end=self.lum[1], halo_taa=5 - self.rank % 2, halo_ssr=7 - self.rank % 2, halo_ssao=11 - self.rank % 2)


def sharded_worker(rank, world, port, height, q, cuts=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diligentfx_amd.sharded import HISTORY_PLANES, ShardedChain, TorchDistComm

        g = torch.Generator().manual_seed(99)
        lvl = height // 8 + 1  # a level height that does not split evenly over the ranks
        full = {"radiance": torch.rand(height, 40, generator=g), "bloom_gather": torch.rand(lvl, 12, generator=g)}
        full["ae_low_res"] = torch.rand(64, 128, generator=g)
        for name, _ in HISTORY_PLANES:
            full[name] = torch.rand(height, 20, generator=g)
        lcuts = [round(i * lvl / world) for i in range(world + 1)]
        chain = _FakeChain(full, rank, world, height, (lcuts[rank], lcuts[rank + 1]), cuts)
        sh = ShardedChain(chain, height, rank, world, 3, cuts)
        sh.step(None, TorchDistComm(rank, world, cuts=cuts))
        b, e = sh.band
        ok = chain.band == (b, e, 3)
        ok = ok and torch.equal(chain.planes["bloom_gather"], full["bloom_gather"])
        ok = ok and torch.equal(chain.seen_luminance, full["ae_low_res"])
        halos = {"taa_history": 5, "ssr_history_radiance": 7, "ssr_history_variance": 7, "ssao_history_ao": 11, "ssao_history_len": 11}
        for name, h in halos.items():
            lo, hi = max(b - h, 0), min(e + h, height)
            p = chain.planes[name]
            ok = ok and torch.equal(p[lo:hi], full[name][lo:hi]) and bool(torch.isnan(p[:lo]).all()) and bool(torch.isnan(p[hi:]).all())
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,height,cuts", [(2, 64, None), (3, 96, None), (3, 96, (0, 40, 60, 96))])
def test_sharded_driver_exchanges_gloo(world, height, cuts):
    """ShardedChain.step over a real process group: the gathers of disjoint rows by summation (Bloom level, luminance rows) and the history halos."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = free_port()
    procs = [ctx.Process(target=sharded_worker, args=(r, world, port, height, q, cuts)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(res) == [(r, True) for r in range(world)]
