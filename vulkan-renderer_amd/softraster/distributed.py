"""Multi-GPU sharding of the draw path, one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Two ways to shard, both new relative to the reference (it is single-device, src/vulkan_helper/gpu_device.cpp:83-102):

* sort-last by view: camera views are independent units; view k goes to rank k % world.  No collective
  on the data path; finished frames can stay resident or be gathered.
* sort-first by screen region: every rank holds the whole mesh and textures, rasterises one horizontal
  band of the frame (a render target restricted to a rectangle) and the finished BGRA8 bands are sent
  to one rank.  Visibility is partition-independent by construction, so the composite equals the
  single-GPU frame bit for bit; no depth exchange is needed because the bands are disjoint.  The cut
  rows can follow a cost profile (balance_bands), so that a frame whose geometry sits in the middle
  rows does not leave the outer ranks idle.  The exchange is point to point (BandCompositor): every
  peer sends its band straight into its rows of the root's frame -- on an xGMI node each peer has its
  own link to the root -- with no padding and no assembly copy, and two frame slots so that the
  exchange of one frame overlaps the drawing of the next.
"""
from __future__ import annotations

import numpy as np


def shard_views(n_views: int, rank: int, world: int) -> list[int]:
    """Indices of the views rank `rank` draws (round robin)."""
    return list(range(rank, n_views, world))


def split_bands(width: int, height: int, world: int, align: int = 64) -> list[tuple[int, int, int, int]]:
    """`world` horizontal bands (x, y, w, h) of (nearly) equal height covering the frame; cut rows are multiples of
    `align` (the tile size) where the height allows, so that no tile is rasterised by two ranks."""
    if world <= 0:
        raise ValueError("world must be positive")
    units = max(1, -(-height // align))
    cuts = [min(height, (units * r // world) * align) for r in range(world + 1)]
    cuts[-1] = height
    if len(set(cuts)) != world + 1:     # too few tile rows: fall back to row granularity
        cuts = [height * r // world for r in range(world + 1)]
    return [(0, cuts[r], width, cuts[r + 1] - cuts[r]) for r in range(world)]


def balance_bands(row_cost, width: int, world: int, align: int = 64) -> list[tuple[int, int, int, int]]:
    """`world` horizontal bands whose summed `row_cost` (one non-negative number per frame row, e.g. a constant per row
    plus the covered pixels of the previous frame's row) is as equal as cut rows at multiples of `align` allow.  Every
    band is at least one row high; the bands cover the frame."""
    cost = np.asarray(row_cost, np.float64)
    height = cost.size
    if world <= 0 or height < world:
        raise ValueError("need at least one row per band")
    cum = np.concatenate([[0.0], np.cumsum(np.maximum(cost, 0.0))])
    total = cum[-1]
    if not total > 0.0:
        return split_bands(width, height, world, align)
    cuts = [0]
    for r in range(1, world):
        y = int(np.searchsorted(cum, total * r / world))            # first row boundary at or above the target share
        snapped = int(round(y / align)) * align
        lo, hi = cuts[-1] + 1, height - (world - r)                  # leave a row for every band still to come
        y = snapped if lo <= snapped <= hi else min(max(y, lo), hi)
        cuts.append(y)
    cuts.append(height)
    return [(0, cuts[r], width, cuts[r + 1] - cuts[r]) for r in range(world)]


class BandCompositor:
    """Collects the bands of a sort-first frame on rank `dst` with point-to-point transfers.

    Every rank renders into ``local(slot)``: on `dst` that is a view of the rows of its band inside the frame of the
    slot, elsewhere a buffer of the band's size.  ``exchange(slot)`` starts the transfers of the slot (peers send,
    `dst` receives straight into the other bands' rows) and returns at once; ``wait(slot)`` blocks the host until they
    are done -- after it the frame of the slot is complete on `dst` (``frame(slot)``) and every rank may render into
    the slot again.  With two slots the exchange of frame i runs while frame i + 1 is drawn.  All tensors are
    allocated once, here.
    """

    def __init__(self, bands, rank: int, world: int, device: str, dst: int = 0, slots: int = 2, group=None):
        import torch
        self.bands, self.rank, self.world, self.dst, self.group = list(bands), rank, world, dst, group
        self.width = bands[0][2]
        self.height = sum(b[3] for b in bands)
        x, y, w, h = bands[rank]
        if rank == dst:
            self._frames = [torch.zeros((self.height, self.width, 4), dtype=torch.uint8, device=device) for _ in range(slots)]
            self._local = [f[y:y + h] for f in self._frames]
        else:
            self._frames = [None] * slots
            self._local = [torch.zeros((h, w, 4), dtype=torch.uint8, device=device) for _ in range(slots)]
        self._pending = [[] for _ in range(slots)]

    def local(self, slot: int):
        return self._local[slot]

    def frame(self, slot: int):
        """The assembled (H, W, 4) frame of the slot on `dst` (None elsewhere); complete after wait(slot)."""
        return self._frames[slot]

    def exchange(self, slot: int):
        import torch.distributed as dist
        if self.world == 1:
            return
        ops = []
        if self.rank == self.dst:
            for r, (x, y, w, h) in enumerate(self.bands):
                if r != self.dst:
                    ops.append(dist.P2POp(dist.irecv, self._frames[slot][y:y + h], self._peer(r), group=self.group))
        else:
            ops.append(dist.P2POp(dist.isend, self._local[slot], self._peer(self.dst), group=self.group))
        self._pending[slot] = dist.batch_isend_irecv(ops)

    def _peer(self, group_rank: int) -> int:
        import torch.distributed as dist
        return dist.get_global_rank(self.group, group_rank) if self.group is not None else group_rank

    def wait(self, slot: int):
        import torch
        for work in self._pending[slot]:
            work.wait()
        self._pending[slot] = []
        if self._local[slot].is_cuda:     # work.wait() only orders the current torch stream after the transfer: make the host wait too,
            torch.cuda.current_stream(self._local[slot].device).synchronize()   # because the renderer draws on a stream of its own


def timed_alone_then_together(timed, rank: int, world: int, alone=None, device: str = "cpu", group=None):
    """The timing protocol of a multi-rank benchmark line that carries its own one-rank point.

    `timed()` runs this rank's share of the timed series and returns its duration in seconds; `alone()` (default:
    `timed`) is what rank 0 runs by itself first -- the same workload on one GPU -- while every other rank waits at a
    barrier and touches nothing.  Then all ranks start `timed()` together from a barrier.  Returns
    (alone_seconds, together_seconds, own_seconds): rank 0's solo duration (broadcast to every rank), the MAX over the
    ranks of the joint run, and this rank's own duration in the joint run.  Tensors for the collectives live on
    `device` ("cuda:<n>" under RCCL, "cpu" under gloo)."""
    import torch
    import torch.distributed as dist
    solo = torch.zeros(1, dtype=torch.float64, device=device)
    if world > 1:
        dist.barrier(group=group)
    if rank == 0:
        solo[0] = float((alone or timed)())
    if world > 1:
        dist.barrier(group=group)             # the others have waited here while rank 0 ran alone
        dist.broadcast(solo, src=_global_rank(group, 0), group=group)
        dist.barrier(group=group)             # everyone starts the joint run from here
    own = float(timed())
    together = torch.tensor([own], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(together, op=dist.ReduceOp.MAX, group=group)
    return float(solo.item()), float(together.item()), own


def _global_rank(group, group_rank: int) -> int:
    import torch.distributed as dist
    return dist.get_global_rank(group, group_rank) if group is not None else group_rank


def gather_bands(local_band, bands, rank: int, dst: int = 0, group=None):
    """One-shot form of the exchange for a single frame: sends the per-rank BGRA8 bands ((h_r, w, 4) uint8 torch tensors
    on the rank's device) to `dst` and returns the assembled (H, W, 4) frame there (None elsewhere)."""
    import torch.distributed as dist
    world = dist.get_world_size(group)
    comp = BandCompositor(bands, rank, world, str(local_band.device), dst=dst, slots=1, group=group)
    comp.local(0).copy_(local_band)
    comp.exchange(0)
    comp.wait(0)
    return comp.frame(0)


def render_band_to_tensor(sr, device, mesh, textures, ubo, width, height, band, samples=1):
    """Draws `band` of the frame on this rank's GPU straight into a torch tensor (the colour target is bound
    to the tensor's memory, so the exchange needs no extra copy).  Returns (tensor, target)."""
    import torch
    x, y, w, h = band
    out = torch.empty((h, w, 4), dtype=torch.uint8, device=f"cuda:{device.ordinal}")
    target = sr.RenderTarget(device, width, height, samples=samples, n_views=1, rect=band)
    target.bind_color(out.data_ptr())
    sr.draw(device, target, mesh, ubo, textures)
    device.synchronize()
    return out, target
