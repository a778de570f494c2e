"""Row-band sharding of a frame across the GPUs of one node -- partition math and the two exchange primitives the sharded
chain needs, on top of torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box, "gloo" in the CPU tests).

Layout: every rank holds FULL-FRAME planes in global coordinates (288 GB of HBM per GPU makes that free: a 8K float4 plane is 531 MB); only
the rows of its band (+ ghost rows) are valid.  Pitched planes make a block of k rows one contiguous slab, so

  * exchange_halos(): each rank sends the first / last `halo` rows of its band to its upper / lower neighbour and receives their
    boundary rows into its ghost rows (grouped isend/irecv = ncclSend/ncclRecv pairs, at most two peers, no ring);
  * allgather_rows(): in-place all-gather of the bands into the full plane (for the unbounded-reach inputs of the SSR ray march:
    Hi-Z, scene colour, normals) -- one hop on the fully connected xGMI topology.

Status: these primitives are covered by world-size-2/3 gloo tests; the kernels do not take a row window yet, so bench.py --gpus N runs
one independent view per GPU (weak scaling, no data-path collective).  See DESIGN.md section 6."""
from dataclasses import dataclass

import torch
import torch.distributed as dist


@dataclass(frozen=True)
class RowBands:
    """Partition of `height` rows into `world` equal bands (height % world == 0 keeps the all-gather in place)."""
    height: int
    world: int
    halo: int = 0

    def __post_init__(self):
        if self.height % self.world != 0:
            raise ValueError(f"height {self.height} is not divisible by the number of ranks {self.world}")
        if self.halo < 0 or self.halo > self.rows:
            raise ValueError(f"halo {self.halo} must lie in [0, band height {self.rows}]")

    @property
    def rows(self):
        return self.height // self.world

    def band(self, rank):
        """[begin, end) rows owned by `rank`."""
        return rank * self.rows, (rank + 1) * self.rows

    def extended(self, rank):
        """[begin, end) rows valid on `rank` after a halo exchange (band + ghost rows, clipped to the frame)."""
        b, e = self.band(rank)
        return max(b - self.halo, 0), min(e + self.halo, self.height)

    def mip(self, level):
        """The same partition on mip `level` (requires band edges aligned to 2^level)."""
        if self.rows % (1 << level) != 0:
            raise ValueError(f"band height {self.rows} is not a multiple of 2^{level}: align the bands or exchange a wider halo")
        return RowBands(self.height >> level, self.world, min(self.halo, self.rows >> level))


def exchange_halos(plane: torch.Tensor, bands: RowBands, rank: int, group=None):
    """plane: (H, W[, C]) full-frame tensor whose rows [band) are valid on this rank; fills the ghost rows from the neighbours.
    Returns the list of completed P2P ops (empty for world == 1 or halo == 0)."""
    if bands.world == 1 or bands.halo == 0:
        return []
    assert plane.shape[0] == bands.height and plane.is_contiguous()
    b, e = bands.band(rank)
    h = bands.halo
    ops = []
    if rank > 0:  # upper neighbour: send my first rows, receive its last rows into my upper ghost zone
        ops.append(dist.P2POp(dist.isend, plane[b:b + h], rank - 1, group))
        ops.append(dist.P2POp(dist.irecv, plane[b - h:b], rank - 1, group))
    if rank < bands.world - 1:
        ops.append(dist.P2POp(dist.isend, plane[e - h:e], rank + 1, group))
        ops.append(dist.P2POp(dist.irecv, plane[e:e + h], rank + 1, group))
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
    flat = plane.view(bands.world, -1)  # bands are equal-sized contiguous slabs
    work = dist.all_gather_into_tensor(flat, plane[b:e].reshape(1, -1).contiguous(), group=group, async_op=async_op)
    return work if async_op else plane


def max_over_ranks(value: float, device, group=None) -> float:
    """Timing reduction used by bench.py: the slowest rank defines the step time."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return float(t.item())
