"""Multi-GPU sharding of the draw path, one process per GPU (torch.distributed; backend "nccl" is RCCL
over xGMI on ROCm, "gloo" on CPU for tests).

Two ways to shard, both new relative to the reference (it is single-device, src/vulkan_helper/gpu_device.cpp:83-102):

* sort-last by view: camera views are independent units; view k goes to rank k % world.  No collective
  on the data path; finished frames can stay resident or be gathered.
* sort-first by screen region: every rank holds the whole mesh and textures, rasterises one horizontal
  band of the frame (a render target restricted to a rectangle) and the finished BGRA8 bands are gathered
  on one rank.  Visibility is partition-independent by construction, so the composite equals the
  single-GPU frame bit for bit; no depth exchange is needed because the bands are disjoint.
"""
from __future__ import annotations

import numpy as np


def shard_views(n_views: int, rank: int, world: int) -> list[int]:
    """Indices of the views rank `rank` draws (round robin)."""
    return list(range(rank, n_views, world))


def split_bands(width: int, height: int, world: int, align: int = 64) -> list[tuple[int, int, int, int]]:
    """`world` horizontal bands (x, y, w, h) covering the frame; cut rows are multiples of `align` (the tile
    size) where the height allows, so that no tile is rasterised by two ranks."""
    if world <= 0:
        raise ValueError("world must be positive")
    units = max(1, -(-height // align))
    cuts = [min(height, (units * r // world) * align) for r in range(world + 1)]
    cuts[-1] = height
    if len(set(cuts)) != world + 1:     # too few tile rows: fall back to row granularity
        cuts = [height * r // world for r in range(world + 1)]
    return [(0, cuts[r], width, cuts[r + 1] - cuts[r]) for r in range(world)]


def gather_bands(local_band, bands, rank: int, dst: int = 0, group=None):
    """Gathers the per-rank BGRA8 bands ((h_r, w, 4) uint8 torch tensors on the rank's device) on `dst` and
    returns the assembled (H, W, 4) frame there (None elsewhere).  One collective: each peer sends its
    band straight to the root, which on an xGMI node uses a different link per peer."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)
    w = bands[0][2]
    hmax = max(b[3] for b in bands)
    send = local_band
    if local_band.shape[0] != hmax:      # gather wants equal shapes: pad the short bands
        send = torch.zeros((hmax, w, 4), dtype=torch.uint8, device=local_band.device)
        send[: local_band.shape[0]] = local_band
    recv = [torch.empty_like(send) for _ in range(world)] if rank == dst else None
    dist.gather(send.contiguous(), recv, dst=dst, group=group)
    if rank != dst:
        return None
    frame = torch.empty((sum(b[3] for b in bands), w, 4), dtype=torch.uint8, device=local_band.device)
    for r, (x, y, bw, bh) in enumerate(bands):
        frame[y:y + bh] = recv[r][:bh]
    return frame


def render_band_to_tensor(sr, device, mesh, textures, ubo, width, height, band, samples=1):
    """Draws `band` of the frame on this rank's GPU straight into a torch tensor (the colour target is bound
    to the tensor's memory, so the collective needs no extra copy).  Returns (tensor, target)."""
    import torch
    x, y, w, h = band
    out = torch.empty((h, w, 4), dtype=torch.uint8, device=f"cuda:{device.ordinal}")
    target = sr.RenderTarget(device, width, height, samples=samples, n_views=1, rect=band)
    target.bind_color(out.data_ptr())
    sr.draw(device, target, mesh, ubo, textures)
    device.synchronize()
    return out, target
