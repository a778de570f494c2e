"""One frame of the chain sharded by row bands over the GPUs of a node.

Each rank owns H / N consecutive rows of the final image and runs mifx_chain_execute_phase on them; every pass is launched on the rows its
consumers need (the band grown by the downstream reach -- redundant compute instead of a halo exchange per pass).  Two exchanges per frame
remain, because their reach is not bounded by a few rows (the third, the all-gather of the shaded radiance for SSR's ray march, is gone since
round 3: the colour at a ray hit outside the rank's own rows is shaded on the spot, api_chain.cpp phase 2):

  after phase 2   Bloom level 1 (1/16 of the pixels): every rank contributes the rows it owns             one all-reduce(sum) of disjoint rows
  after phase 3   history planes (TAA, SSR radiance / variance, SSAO AO / length): ghost rows <- neighbours   grouped send / recv, <= 2 peers
  auto exposure   phase 3 then ends with this rank's rows of the 64 x 64 low-resolution luminance; all ranks get all rows   one all-reduce(sum), 32 KB
                  (rows are disjoint), phase 4 reduces them -- the same values in the same order on every rank -- and tone-maps the band

The exchanges are written against a small communicator interface so that the same driver runs over torch.distributed (backend "nccl" = RCCL
on the GPU box, "gloo" in the CPU tests of the primitives) and over an in-process emulation of N ranks on one GPU (tests/test_gpu_sharded.py:
the sharded result must equal the unsharded one bit for bit)."""
import torch
import torch.distributed as dist

from . import dist as D

HISTORY_PLANES = (("taa_history", "halo_taa"), ("ssr_history_radiance", "halo_ssr"), ("ssr_history_variance", "halo_ssr"),
                  ("ssao_history_ao", "halo_ssao"), ("ssao_history_len", "halo_ssao"))


class TorchDistComm:
    """The exchanges over a torch.distributed process group (one rank per GPU). cuts: row boundaries of uneven bands (None: equal)."""

    def __init__(self, rank, world, group=None, cuts=None):
        self.rank, self.world, self.group, self.cuts = rank, world, group, cuts

    def bands(self, height, halo=0):
        return D.RowBands(height, self.world, halo, self.cuts)

    def allgather_rows(self, plane, height, async_op=False):
        return D.allgather_rows(plane, self.bands(height), self.rank, self.group, async_op=async_op)

    def max_over_ranks(self, values, device):
        """Element-wise maximum of a short list of ints over the ranks (the halo sizes must be the same on both sides of an exchange)."""
        t = torch.tensor(list(values), dtype=torch.int64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return [int(v) for v in t.tolist()]

    def gather_owned_rows(self, plane, own_begin, own_end):
        # rows are owned by exactly one rank: zero the others and sum (x + 0 == x exactly), no equal-split constraint on the level height
        plane[:own_begin].zero_()
        plane[own_end:].zero_()
        dist.all_reduce(plane, op=dist.ReduceOp.SUM, group=self.group)

    def exchange_halos(self, plane, height, halo):
        D.exchange_halos(plane, self.bands(height, halo), self.rank, self.group)


class ShardedChain:
    """Drives one rank: chain = api.Chain with its inputs bound per frame; comm = TorchDistComm (or the emulation in the tests)."""

    def __init__(self, chain, height, rank, world, max_motion_rows, cuts=None):
        """cuts: world + 1 row boundaries for bands of unequal height (e.g. weighted by cost); None: equal bands (height % world == 0)."""
        self.chain, self.height, self.rank, self.world = chain, height, rank, world
        self.bands = D.RowBands(height, world, 0, tuple(cuts) if cuts is not None else None)
        self.band = self.bands.band(rank)
        self.halos = None  # common to all ranks, agreed on at every history exchange
        chain.set_row_band(self.band[0], self.band[1], max_motion_rows)

    def phase(self, bound, k):
        self.chain.execute_phase(bound, k)

    PHASES = 5  # (phase 4 does nothing unless auto exposure is on)

    def exchange(self, bound, k, comm, async_op=False):
        """The exchange that follows phase k (phase 1 has none; phase 3: the luminance rows when auto exposure is on; the history halos follow the
        last phase)."""
        c = self.chain
        if k == 3:
            if getattr(c, "auto_exposure", False):
                info = c.shard_info(bound)
                comm.gather_owned_rows(c.shard_plane("ae_low_res"), info.ae_begin, info.ae_end)
            return None
        if k in (0, 1):
            return None
        if k == 2:
            info = c.shard_info(bound)
            if info.gather_level >= 0:
                comm.gather_owned_rows(c.shard_plane("bloom_gather"), info.own_begin, info.own_end)
        else:
            # every rank derives its own halo need (window ghost + motion bound); both sides of an exchange must move the same rows.  Agreed on every
            # frame: the needs follow per-frame attributes (SSAO SpatialReconstructionRadius, Bloom Radius -> mip count -> the TAA window), and one
            # all-reduce of three integers is noise beside the exchanges themselves.
            info = c.shard_info(bound)
            fields = sorted({f for _, f in HISTORY_PLANES})
            agreed = comm.max_over_ranks([getattr(info, f) for f in fields], c.device)
            self.halos = dict(zip(fields, agreed))
            for name, field in HISTORY_PLANES:
                comm.exchange_halos(c.shard_plane(name), self.height, self.halos[field])

        return None

    def step(self, bound, comm):
        self.phase(bound, 0)
        self.phase(bound, 1)
        self.phase(bound, 2)
        self.exchange(bound, 2, comm)
        self.phase(bound, 3)
        self.exchange(bound, 3, comm)
        self.phase(bound, 4)
        self.exchange(bound, 4, comm)
