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


@pytest.mark.parametrize("world,height,halo", [(2, 64, 8), (3, 96, 5), (2, 32, 16)])
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
