"""Row-band sharding of a frame across the GPUs of one node -- partition math and the two exchange primitives the sharded
chain needs, on top of torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Layout: every rank holds FULL-FRAME planes in global coordinates (288 GB of HBM per GPU makes that free: a 8K float4 plane is 531 MB); only
the rows of its band (+ ghost rows) are valid.  Pitched planes make a block of k rows one contiguous slab, so

  * exchange_halos(): each rank sends the first / last `halo` rows of its band to its upper / lower neighbour and receives their
    boundary rows into its ghost rows (grouped isend/irecv = ncclSend/ncclRecv pairs, at most two peers, no ring);
  * allgather_rows(): in-place all-gather of the bands into the full plane (for the unbounded-reach inputs of the SSR ray march:
    Hi-Z, scene colour, normals) -- one hop on the fully connected xGMI topology.

Status: these primitives carry the torch.distributed variant of the sharded chain (sharded.py: the path the world-size-2/3 gloo tests and
`bench.py --comm torch` run); the default for `bench.py --gpus N` is the same exchange pattern inside the library (csrc/api_comm.cpp:
mifx_chain_execute_sharded, grouped ncclSend / ncclRecv).  See DESIGN.md section 6."""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class RowBands:
    """Partition of `height` rows into `world` bands: equal by default (height % world == 0 keeps the all-gather in place), or at the row
    boundaries `cuts` (world + 1 increasing values from 0 to height) for cost-weighted bands."""
    height: int
    world: int
    halo: int = 0
    cuts: tuple = None

    def __post_init__(self):
        if self.cuts is None:
            if self.height % self.world != 0:
                raise ValueError(f"height {self.height} is not divisible by the number of ranks {self.world}")
        else:
            c = tuple(int(v) for v in self.cuts)
            if len(c) != self.world + 1 or c[0] != 0 or c[-1] != self.height or any(c[i] >= c[i + 1] for i in range(self.world)):
                raise ValueError(f"cuts {c} do not partition {self.height} rows into {self.world} bands")
            object.__setattr__(self, "cuts", c)
        if self.halo < 0 or self.halo > self.height:
            raise ValueError(f"halo {self.halo} must lie in [0, frame height {self.height}]")

    @property
    def equal(self):
        return self.cuts is None

    @property
    def rows(self):
        """Height of the (smallest) band."""
        if self.cuts is None:
            return self.height // self.world
        return min(self.cuts[i + 1] - self.cuts[i] for i in range(self.world))

    @property
    def max_rows(self):
        if self.cuts is None:
            return self.height // self.world
        return max(self.cuts[i + 1] - self.cuts[i] for i in range(self.world))

    def band(self, rank):
        """[begin, end) rows owned by `rank`."""
        if self.cuts is None:
            return rank * self.rows, (rank + 1) * self.rows
        return self.cuts[rank], self.cuts[rank + 1]

    def extended(self, rank):
        """[begin, end) rows valid on `rank` after a halo exchange (band + ghost rows, clipped to the frame)."""
        b, e = self.band(rank)
        return max(b - self.halo, 0), min(e + self.halo, self.height)

    def mip(self, level):
        """The same partition on mip `level` (requires band edges aligned to 2^level)."""
        if self.cuts is not None:
            raise ValueError("mip() is defined for equal bands")
        if self.rows % (1 << level) != 0:
            raise ValueError(f"band height {self.rows} is not a multiple of 2^{level}: align the bands or exchange a wider halo")
        return RowBands(self.height >> level, self.world, min(self.halo, self.rows >> level))


def halo_rows(bands: RowBands, owner: int, taker: int):
    """[begin, end) rows of `owner`'s band that lie in the ghost zone of `taker` (the `halo` rows above taker's band when owner is above it, below otherwise)."""
    tb, te = bands.band(taker)
    gb, ge = (max(tb - bands.halo, 0), tb) if owner < taker else (te, min(te + bands.halo, bands.height))
    ob, oe = bands.band(owner)
    return max(ob, gb), min(oe, ge)


def exchange_halos(plane: torch.Tensor, bands: RowBands, rank: int, group=None):
    """plane: (H, W[, C]) full-frame tensor whose rows [band) are valid on this rank; fills the ghost rows (`halo` rows above and below the band) from the
    ranks that own them -- the neighbours, and the ranks beyond them when the halo is taller than a neighbour's band.
    Returns the list of completed P2P ops (empty for world == 1 or halo == 0)."""
    if bands.world == 1 or bands.halo == 0:
        return []
    assert plane.shape[0] == bands.height and plane.is_contiguous()
    ops = []
    for q in range(bands.world):
        if q == rank:
            continue
        out, inn = halo_rows(bands, rank, q), halo_rows(bands, q, rank)  # rows of mine that q's ghost zone wants / rows of q's that mine wants
        if out[1] > out[0]:
            ops.append(dist.P2POp(dist.isend, plane[out[0]:out[1]], q, group))
        if inn[1] > inn[0]:
            ops.append(dist.P2POp(dist.irecv, plane[inn[0]:inn[1]], q, group))
    if not ops:
        return []
    reqs = dist.batch_isend_irecv(ops)
    for r in reqs:
        r.wait()
    return reqs


def allgather_rows(plane: torch.Tensor, bands: RowBands, rank: int, group=None, async_op=False):
    """In-place all-gather: after the call (after .wait() on the returned work for async_op=True) every rank holds all rows of `plane`
    (each rank contributed its band)."""
    if bands.world == 1:
        return None if async_op else plane
    assert plane.shape[0] == bands.height and plane.is_contiguous()
    b, e = bands.band(rank)
    if bands.equal:
        flat = plane.view(bands.world, -1)  # bands are equal-sized contiguous slabs: gathered in place
        work = dist.all_gather_into_tensor(flat, plane[b:e].reshape(1, -1).contiguous(), group=group, async_op=async_op)
        return work if async_op else plane
    # uneven bands: gather slabs padded to the tallest band, then copy every rank's rows into place (two local copies of the plane)
    row = plane[0].numel()
    stage_in = plane.new_zeros(bands.max_rows, row)
    stage_in[: e - b].copy_(plane[b:e].reshape(e - b, row))
    flat_out = plane.new_empty(bands.world * bands.max_rows, row)  # concatenation along dim 0 (the layout every backend accepts)
    stage_out = flat_out.view(bands.world, bands.max_rows, row)
    work = dist.all_gather_into_tensor(flat_out, stage_in, group=group, async_op=async_op)

    def scatter():
        for r in range(bands.world):
            rb, re_ = bands.band(r)
            if r != rank:
                plane[rb:re_].reshape(re_ - rb, row).copy_(stage_out[r, : re_ - rb])

    if async_op:
        return _DeferredWork(work, scatter)
    scatter()
    return plane


class _DeferredWork:
    """An asynchronous collective followed by local copies: wait() orders both before the caller's stream continues."""

    def __init__(self, work, after):
        self.work, self.after = work, after

    def wait(self):
        self.work.wait()
        self.after()


def max_over_ranks(value: float, device, group=None) -> float:
    """Timing reduction used by bench.py: the slowest rank defines the step time."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
