"""The N > 1 paths on CPU: gloo process groups of 2 and 4 ranks.  The renderer of each rank is the CPU
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
        elif mode == "balanced_pipelined":
            # cost-balanced bands from the first frame's coverage, then three frames of a moving camera through the
            # two-slot compositor: the exchange of a frame is still in flight while the next one is rendered
            ubos = scene.orbit_ubos(3, W, H)
            even = d.split_bands(W, H, world, align=8)
            mine = oracle.draw(ubos[0], verts, idx, tex, W, H, rect=even[rank], shade=False)
            rows = torch.zeros(H, dtype=torch.float64)
            rows[even[rank][1]:even[rank][1] + even[rank][3]] = torch.from_numpy((mine["tri"][..., 0] != 0xFFFFFFFF).sum(axis=1).astype(np.float64))
            dist.all_reduce(rows)           # every rank learns the whole profile and derives the same bands
            bands = d.balance_bands(W / 8 + rows.numpy(), W, world, align=8)
            comp = d.BandCompositor(bands, rank, world, "cpu", dst=0, slots=2)
            ok = sum(b[3] for b in bands) == H and all(b[3] > 0 for b in bands)
            frames = []
            for f in range(3):
                slot = f % 2
                comp.wait(slot)             # the slot's previous exchange (frame f - 2) is over: it may be drawn into again
                if rank == 0 and f >= 2:
                    frames.append(comp.frame(slot).numpy().copy())
                comp.local(slot).copy_(torch.from_numpy(oracle.draw(ubos[f], verts, idx, tex, W, H, rect=bands[rank])["bgra"]))
                comp.exchange(slot)
            for f in (1, 2):
                comp.wait(f % 2)
                if rank == 0:
                    frames.append(comp.frame(f % 2).numpy().copy())
            if rank == 0:
                for f in range(3):
                    ok = ok and bool(np.array_equal(frames[f], oracle.draw(ubos[f], verts, idx, tex, W, H)["bgra"]))
                q.put(ok)
        elif mode == "alone_then_together":
            # the timing protocol of the multi-GPU benchmark line: rank 0 runs the workload by itself while the others wait,
            # then every rank runs its share from a common barrier
            import time
            windows = []

            def timed():
                t0 = time.monotonic(); time.sleep(0.15); t1 = time.monotonic()
                windows.append((t0, t1))
                return t1 - t0
            alone_s, together_s, own_s = d.timed_alone_then_together(timed, rank, world)
            every = [None] * world
            dist.all_gather_object(every, (windows, alone_s, together_s, own_s))
            if rank == 0:
                ok = len(every[0][0]) == 2 and all(len(w[0]) == 1 for w in every[1:])          # rank 0 ran twice, the others once
                solo = every[0][0][0]
                joint = [every[0][0][1]] + [w[0][0] for w in every[1:]]
                ok = ok and all(solo[1] <= j[0] for j in joint)                                   # nothing ran beside the solo run
                ok = ok and max(j[0] for j in joint) < min(j[1] for j in joint)                  # the joint runs overlap: they started together
                ok = ok and all(abs(e[1] - (solo[1] - solo[0])) < 1e-9 for e in every)            # every rank knows the solo time
                ok = ok and all(abs(e[2] - max(x[3] for x in every)) < 1e-9 for e in every)       # and the max over the ranks
                q.put(bool(ok))
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


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("mode", ["sort_first", "sort_last", "balanced_pipelined", "alone_then_together"])
def test_ranks(mode, world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000) + ["sort_first", "sort_last", "balanced_pipelined", "alone_then_together"].index(mode) * 2 + world // 4
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
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


def test_balanced_bands():
    _setup_paths()
    from softraster import distributed as d
    H, W = 4320, 7680
    rows = np.zeros(H)
    rows[590:3730] = np.sqrt(np.maximum(0.0, 1570.0 ** 2 - (np.arange(590, 3730) - 2160.0) ** 2)) * 2     # chord lengths of the benchmark's disc
    cost = W / 8 + rows
    for n in (2, 4, 8):
        bands = d.balance_bands(cost, W, n)
        assert len(bands) == n and bands[0][1] == 0 and sum(b[3] for b in bands) == H and all(b[3] > 0 for b in bands)
        assert all(bands[r][1] + bands[r][3] == bands[r + 1][1] for r in range(n - 1)) and all(b[1] % 64 == 0 for b in bands)
        shares = [cost[b[1]:b[1] + b[3]].sum() / cost.sum() for b in bands]
        even = [cost[b[1]:b[1] + b[3]].sum() / cost.sum() for b in d.split_bands(W, H, n)]
        assert max(shares) < 1.15 / n, (n, shares)
        if n == 8:
            assert max(even) > 1.5 / n     # what equal heights would give: the middle ranks carry the disc
    assert d.balance_bands(np.zeros(100), 64, 4) == d.split_bands(64, 100, 4)
    few = d.balance_bands([0, 0, 0, 5.0], 16, 4, align=64)
    assert [b[3] for b in few] == [1, 1, 1, 1]
