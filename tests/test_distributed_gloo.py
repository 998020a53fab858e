"""The N > 1 paths on CPU: world_size-2 gloo process groups.  The renderer of each rank is the CPU
oracle here (there is no GPU in this environment); what is under test is the host-side sharding and
compositing of softraster.distributed: bands tile the frame, the gathered composite is bit-identical
to the single-process frame, views are dealt round robin."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _setup_paths():
    for p in (ROOT, os.path.join(ROOT, "vulkan-renderer_amd"), os.path.join(ROOT, "tests")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _worker(rank, world, port, mode, q):
    _setup_paths()
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from softraster import scene, distributed as d
        from oracle import oracle
        W, H = 160, 200
        verts, idx = scene.uv_sphere(40, 30)
        tex = scene.material_textures(32)
        if mode == "sort_first":
            ubo = scene.default_camera_ubo(W, H)
            bands = d.split_bands(W, H, world)
            mine = oracle.draw(ubo, verts, idx, tex, W, H, samples=1, rect=bands[rank])
            frame = d.gather_bands(torch.from_numpy(mine["bgra"]), bands, rank, dst=0)
            if rank == 0:
                full = oracle.draw(ubo, verts, idx, tex, W, H, samples=1)
                q.put(bool(np.array_equal(frame.numpy(), full["bgra"])))
        else:   # sort-last by view
            n_views = 5
            ubos = scene.orbit_ubos(n_views, W, H)
            mine = d.shard_views(n_views, rank, world)
            sums = torch.zeros(n_views, dtype=torch.int64)
            for k in mine:
                sums[k] = int(oracle.draw(ubos[k], verts, idx, tex, W, H)["bgra"].astype(np.int64).sum())
            dist.all_reduce(sums)          # test-only check that every view was drawn exactly once
            if rank == 0:
                want = [int(oracle.draw(ubos[k], verts, idx, tex, W, H)["bgra"].astype(np.int64).sum()) for k in range(n_views)]
                q.put(sums.tolist() == want)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["sort_first", "sort_last"])
def test_two_ranks(mode):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + (0 if mode == "sort_first" else 1)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, mode, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert q.get(timeout=10) is True


def test_band_and_view_partitions():
    _setup_paths()
    from softraster import distributed as d
    for (W, H, n) in [(7680, 4320, 8), (7680, 4320, 4), (3840, 2160, 2), (800, 600, 3), (64, 10, 4), (1920, 1080, 1)]:
        bands = d.split_bands(W, H, n)
        assert len(bands) == n and bands[0][1] == 0 and sum(b[3] for b in bands) == H
        assert all(b[0] == 0 and b[2] == W and b[3] > 0 for b in bands)
        assert all(bands[r][1] + bands[r][3] == bands[r + 1][1] for r in range(n - 1))
    b8 = d.split_bands(7680, 4320, 8)
    assert all(b[1] % 64 == 0 for b in b8) and max(b[3] for b in b8) - min(b[3] for b in b8) <= 64
    assert sorted(sum((d.shard_views(1024, r, 8) for r in range(8)), [])) == list(range(1024))
    assert d.shard_views(5, 1, 2) == [1, 3] and d.shard_views(3, 7, 8) == []
