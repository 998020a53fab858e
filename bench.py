#!/usr/bin/env python3
"""Benchmark of the draw path.

    python bench.py --gpus N --steps K --warmup W

N = 1 (default): BASELINE config 3 -- one 3840x2160 frame of a 10 M-triangle UV sphere with a 4096^2
mip-mapped albedo map (2048^2 metallic / roughness), the reference's default camera, trilinear samplers
(the sampling the headline configuration names; the time with the reference's own 16x anisotropic
sampler is reported beside it).  A step is one sr_draw (clear, vertex, set-up + binning, raster,
fragment) with mesh and textures already resident in HBM.

N > 1: BASELINE config 5 -- the multi-camera batch of a 50 M-triangle mesh sharded sort-last by view:
every rank draws its 128 cameras of a 1024-camera orbit at 1920x1080 per step, no collective on the data
path (weak scaling: N = 8 is the whole batch).  One process per GPU; launched by torch.distributed.run,
or, when WORLD_SIZE is not set, by this script itself, which starts N child ranks before anything touches
a GPU.  --mode sortfirst times BASELINE config 4 instead (one 7680x4320 frame split into bands over the
ranks, bands sent to rank 0 over RCCL).  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import hashlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vulkan-renderer_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

# Set only by tests/emu/rehearse.py: the multi-rank code paths of this script run end to end on CPUs -- ranks talk over gloo, the
# collectives' tensors live in host memory, every rank opens "device 0" of the CPU build of the kernels.  Never set on a GPU box.
CPU_REHEARSAL = os.environ.get("SR_BENCH_CPU_REHEARSAL") == "1"


def tensor_device(local_rank):
    return "cpu" if CPU_REHEARSAL else f"cuda:{local_rank}"


def gpu_synchronize():
    if not CPU_REHEARSAL:
        import torch
        torch.cuda.synchronize()

WORKLOADS = {
    # name: (width, height, sphere X, sphere Y, albedo size, other maps size, samples, views per GPU, cameras of the whole orbit (0: per-rank camera))
    "c3_4k_10Mtri": (3840, 2160, 2500, 2000, 4096, 2048, 1, 1, 0),          # the headline: BASELINE config 3
    "small_1080p_124ktri": (1920, 1080, 250, 248, 1024, 512, 1, 1, 0),       # for quick checks only
    "c1_800x600_msaa4": (800, 600, 88, 88, 2048, 2048, 4, 1, 0),             # the reference's own frame: large triangles, 4 samples
    "c2_1080p_64views": (1920, 1080, 88, 88, 2048, 2048, 1, 64, 0),          # BASELINE config 2: one draw of 64 cameras
    "c5_50Mtri_views": (1920, 1080, 5000, 5000, 2048, 2048, 1, 128, 1024),   # BASELINE config 5: this GPU's 128 of 1024 cameras, 50 M triangles
    "c4_8k_10Mtri": (7680, 4320, 2500, 2000, 4096, 2048, 1, 1, 0),           # BASELINE config 4's frame whole, on one GPU (--mode sortfirst splits it over the ranks)
    "mid_4k_8views": (3840, 2160, 300, 300, 2048, 2048, 1, 8, 0),            # 180 k triangles of some fifty pixels, 8 views: everything takes the tile lists
    "c5_small": (480, 270, 300, 300, 256, 128, 1, 8, 64),                    # config 5's code path at a size a CPU rehearsal can run
}
STAGES = ("clear", "vertex", "setup", "bin", "scan", "fill", "raster", "fragment")


def kernel_bytes(st, n_tiles):
    """Algorithmic (compulsory) HBM bytes of each kernel for one step: every array counted once per
    kernel that must touch it, caches assumed perfect inside a kernel.  st: counters of the draw.
    The fragment figure is the measurement plan's own: visibility read, indices and three 32-byte vertex
    records per distinct visible triangle (capped by the vertex count), unique texels, colour written."""
    V, T, P, S = st["vertices"], st["triangles"], st["pixels"], st.get("samples", 1)
    B, R, Fd, F, Pv = st["triangles_binned"], st["tile_refs"], st["triangles_direct"], st["visible_triangles"], st["shaded_pixels"]
    X = Pv * 3 * 1.33                       # unique texels over the three maps (model)
    direct_samples = Pv * S * (Fd / max(F, 1)) if F else 0
    # V and T are summed over the views; a pass reads positions (vertex kernel) and, when its workgroups take every view of the pass
    # (a mesh large enough to fill the chip that way), indices (set-up kernel) once for all its views
    views, passes = max(st.get("views", 1), 1), max(st.get("passes", 1), 1)
    per_pass = -(-views // passes)

    def reads_per_view(units_per_view, block_units, fused=False, cap=None):
        """How often, per view, a kernel whose workgroups take several views of a pass reads the mesh (the launchers' rule: all views
        of the pass per workgroup unless that leaves fewer than 4096 workgroups; the batch vertex kernel takes at most 8)."""
        grid_x = -(-units_per_view // block_units)
        per_block = per_pass if fused else min(max(per_pass * grid_x // 4096, 1), per_pass, cap or per_pass)
        return passes * -(-per_pass // per_block) / views
    vertex_reads = reads_per_view(V // views, 512, fused=views <= 8, cap=8)
    index_reads = reads_per_view(T // views, 768)
    return {
        "clear": Pv * S * 8,                                                  # visibility words that held something and have to be cleared (the covered pixels' words; the
                                                                              # 32 x 8-pixel blocks the kernel clears hold a few more)
        "vertex": V * 12 * vertex_reads + V * 12,                              # positions read, xy + z planes written
        "setup": T * 12 * index_reads + V * 8 + min(V, 3 * Fd) * 4 + direct_samples * 8 + B * 4,   # indices, xy plane, z of covered micro-triangles, atomic-min words, pending ids
        "bin": B * 4 + B * 12 + min(V, 3 * B) * 8 + R * 12,                   # pending ids, indices, xy, records
        "scan": n_tiles * 12,
        "fill": R * 12 + R * 4,                                                # records read, references written
        "raster": R * 4 + R * 12 + min(V, 3 * R) * 12 + (Pv * S - direct_samples) * 8,
        "fragment": P * S * 8 + F * 12 + min(V, 3 * F) * 32 + X * 4 + P * 4,  # visibility, indices, vertex records, texels, colour
    }


def frame_bytes_reference_model(st):
    """The per-frame figure of the measurement plan this benchmark follows: vertex V*28; set-up/bin
    T*12 + V*16 + R*4; raster R*4 + R*12 + V*16 + P*S*8; fragment P*S*8 + F*108 + X*4 + P*4, with R =
    surviving (triangle, tile) references (here: tile references + directly resolved triangles)."""
    V, T, P, S = st["vertices"], st["triangles"], st["pixels"], st.get("samples", 1)
    R = st["tile_refs"] + st["triangles_direct"]
    F, Pv = st["visible_triangles"], st["shaded_pixels"]
    X = Pv * 3 * 1.33
    return (V * 28) + (T * 12 + V * 16 + R * 4) + (R * 4 + R * 12 + V * 16 + P * S * 8) + (P * S * 8 + F * 108 + X * 4 + P * 4)


def kernel_source_digest():
    """Identifies the kernels a profile was taken on: a digest of the device sources (there is no git on the GPU box)."""
    h = hashlib.sha256()
    for f in ("sr_kernels.hip", "sr_device.h", "sr_launch.h", "sr_api.hip"):
        with open(os.path.join(ROOT, "vulkan-renderer_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(workload, kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/traffic.json, written by
    tools/profiles_to_repo.py) -- only if they were taken on the kernels this run uses; otherwise None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        with open(path) as fh:
            t = json.load(fh)
    except (OSError, ValueError):
        return None
    if t.get("kernel_source_digest") != kernel_source_digest():
        return None
    return t.get(workload, {}).get(kernel)


def lavapipe_status():
    """The intended CPU baseline is the reference itself on Mesa lavapipe.  Probe for a Vulkan loader and an lvp
    ICD; the reference could not be built even if they were there (no GLFW/GLM/Assimp/Conan), so this is
    only reported, never used."""
    import ctypes
    import glob
    try:
        ctypes.CDLL("libvulkan.so.1")
        loader = True
    except OSError:
        loader = False
    icds = [p for d in ("/usr/share/vulkan/icd.d", "/etc/vulkan/icd.d", "/usr/local/share/vulkan/icd.d") for p in glob.glob(os.path.join(d, "lvp_icd*.json"))]
    if not loader and not icds:
        return "lavapipe unavailable (no Vulkan loader, no lvp ICD)"
    return f"Vulkan loader {'present' if loader else 'absent'}, lvp ICD {'present' if icds else 'absent'}; the reference application is not buildable here"


def spawn_ranks(n):
    """--gpus N without a launcher: start N ranks of this script (fresh processes; this one has not touched a GPU and
    does not), wait for them, exit with the worst of their exit codes."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # a rank that dies (no such GPU, out of memory) would leave the others waiting in the rendezvous or a collective until a
    # time-out: stop them as soon as one has failed
    codes = [None] * n
    while any(c is None for c in codes):
        for i, p in enumerate(procs):
            if codes[i] is None:
                codes[i] = p.poll()
        if any(c not in (None, 0) for c in codes):
            failed = next(abs(c) for c in codes if c not in (None, 0))
            for i, p in enumerate(procs):
                if codes[i] is None:
                    p.terminate()
            for i, p in enumerate(procs):
                if codes[i] is None:
                    try:
                        codes[i] = p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill(); codes[i] = p.wait()
            raise SystemExit(failed)   # the code of the rank that failed, not of the ones stopped because of it
        time.sleep(0.2)
    raise SystemExit(max(abs(c) for c in codes))


def init_ranks(args):
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        import torch
        import torch.distributed as dist
        if CPU_REHEARSAL:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        if dist.get_world_size() != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but the process group has {dist.get_world_size()} ranks")
    return rank, local_rank, world


def sort_first(args, sr, rank, local_rank, world):
    """BASELINE config 4: the 10 M-triangle scene at 7680x4320, the frame split into horizontal bands over the ranks
    (mesh and textures replicated; cut rows balanced by the coverage of a first frame), every rank's finished BGRA8
    band sent straight into its rows of rank 0's frame.  Two frame slots: the exchange of a frame overlaps the
    drawing of the next."""
    import torch
    import torch.distributed as dist
    from softraster import distributed as d
    W, H, SX, SY, alb_size, map_size = WORKLOADS[args.workload or "c4_8k_10Mtri"][:6]
    verts, idx = sr.scene.uv_sphere(SX, SY)
    images = [sr.scene.albedo_pattern(alb_size), None, sr.scene.gray_map(map_size, map_size, 0, 254), sr.scene.gray_map(map_size, map_size, 72, 255)]
    ubo = sr.scene.default_camera_ubo(W, H)
    if not CPU_REHEARSAL:
        torch.cuda.set_device(local_rank)
    dev = sr.Device(0 if CPU_REHEARSAL else local_rank)
    mesh = sr.create_mesh_from_data(dev, verts, idx)
    tex = [sr.create_texture(dev, im, sampler=sr.default_sampler(1.0)) if im is not None else None for im in images]
    # cost profile: one frame over equal bands, covered pixels per row, summed over the ranks
    even = d.split_bands(W, H, world)
    probe = sr.RenderTarget(dev, W, H, samples=1, n_views=1, rect=even[rank])
    sr.draw(dev, probe, mesh, ubo, tex)
    rows = torch.zeros(H, dtype=torch.float64, device=tensor_device(local_rank))
    covered = (probe.read_triangle_ids(0)[..., 0] != sr.NO_TRIANGLE).sum(axis=1).astype(np.float64)
    rows[even[rank][1]:even[rank][1] + even[rank][3]] = torch.from_numpy(covered).to(rows.device)
    probe.destroy()
    if world > 1:
        dist.all_reduce(rows)
    bands = d.balance_bands(W / 32 + rows.cpu().numpy(), W, world)   # (a row that holds nothing costs about as much as 240 covered pixels: tools/c4_band_times.py)
    comp = d.BandCompositor(bands, rank, world, tensor_device(local_rank), dst=0, slots=2)
    targets = []
    for slot in range(2):
        t = sr.RenderTarget(dev, W, H, samples=1, n_views=1, rect=bands[rank])
        t.bind_color(comp.local(slot).data_ptr())
        targets.append(t)

    def step(i):
        slot = i % 2
        comp.wait(slot)                  # the slot's previous exchange is over: it may be drawn into again
        sr.draw(dev, targets[slot], mesh, ubo, tex)
        dev.synchronize()                # the band is finished before it is sent
        comp.exchange(slot)

    def drain():
        comp.wait(0); comp.wait(1)
        gpu_synchronize()
        if world > 1:
            dist.barrier()

    for i in range(max(args.warmup, 1)):
        step(i)
    drain()

    def timed():
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        comp.wait(0); comp.wait(1)
        gpu_synchronize()
        return time.perf_counter() - t0

    def alone():
        """The same frame whole on rank 0's GPU (no bands, no exchange): the one-GPU point of this strong-scaling series."""
        whole = sr.RenderTarget(dev, W, H, samples=1, n_views=1)
        for _ in range(max(args.warmup, 1)):
            sr.draw(dev, whole, mesh, ubo, tex)
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sr.draw(dev, whole, mesh, ubo, tex)
        dev.synchronize()
        dt = time.perf_counter() - t0
        whole.destroy()
        return dt
    alone_s, elapsed, _own = d.timed_alone_then_together(timed, rank, world, alone=alone, device=tensor_device(local_rank))
    if rank == 0:
        T = idx.size // 3
        n1 = T * args.steps / alone_s / 1e6
        print(json.dumps({
            "metric": f"Mtris/s, {W}x{H} frame of a {T / 1e6:g}M-triangle mesh split sort-first over the GPUs", "value": round(T * args.steps / elapsed / 1e6, 2),
            "unit": "Mtris/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c4_8k_sort_first" if (args.workload or "c4_8k_10Mtri") == "c4_8k_10Mtri" else args.workload + "_sort_first", "width": W, "height": H, "triangles": T, "bands": bands,
                       "composite": "point-to-point sends of BGRA8 bands into rank 0's frame (RCCL), double-buffered" if world > 1 else "none"},
            "frames_per_s": round(args.steps / elapsed, 2),
            "n1_same_workload_Mtris_per_s": round(n1, 2), "n1_same_workload_ms_per_step": round(alone_s / args.steps * 1e3, 4),
            "speedup_vs_n1_same_workload": round(T * args.steps / elapsed / 1e6 / n1, 3),
            "n1_note": f"rank 0 drew the whole {W}x{H} frame alone, before the joint run, while the other ranks waited at a barrier"}), flush=True)
    drain()
    for t_ in tex:
        if t_ is not None:
            t_.destroy()
    for t_ in targets:
        t_.destroy()
    mesh.destroy(); dev.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c3_4k_10Mtri on one GPU, c5_50Mtri_views (sharded by view) on several")
    ap.add_argument("--mode", default="views", choices=["views", "sortfirst"],
                    help="views: every rank draws its own cameras of the workload (weak scaling, no collective); "
                         "sortfirst: one 7680x4320 frame split into bands over the ranks, bands sent to rank 0 (strong scaling)")
    ap.add_argument("--max-anisotropy", type=float, default=1.0, help="maxAnisotropy of the samplers of the timed run (1: trilinear; the reference's own sampler has 16)")
    ap.add_argument("--lib", default=None, help="time another build of the library (tools/exp_build.py) instead of the shipped one")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true", help="only the timed headline loop and the per-stage pass: no moving-camera figures, no 16x-anisotropic "
                                                                  "second figure, no CPU baseline (profiling runs: every launch in the trace is a launch of the headline frame)")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        spawn_ranks(args.gpus)
    rank, local_rank, world = init_ranks(args)
    if world > 1:
        import torch
        import torch.distributed as dist

    import softraster as sr
    if args.lib:
        sr.LIB_PATH = os.path.abspath(args.lib)
    if args.mode == "sortfirst":
        return sort_first(args, sr, rank, local_rank, world)
    workload = args.workload or ("c3_4k_10Mtri" if world == 1 else "c5_50Mtri_views")
    W, H, SX, SY, alb_size, map_size, samples, views, orbit = WORKLOADS[workload]
    verts, idx = sr.scene.uv_sphere(SX, SY)
    images = [sr.scene.albedo_pattern(alb_size), None, sr.scene.gray_map(map_size, map_size, 0, 254), sr.scene.gray_map(map_size, map_size, 72, 255)]
    if orbit:       # a share of a fixed orbit of cameras: rank r draws cameras r*views .. r*views + views - 1
        ubo = sr.scene.orbit_ubos(views, W, H, first=(rank * views) % orbit, total=orbit)
    elif views > 1:  # a batch of this rank's own
        ubo = sr.scene.orbit_ubos(views, W, H, first=rank * views, total=world * views)
    else:            # rank r draws view r of a `world`-camera orbit; view 0 is the reference's default camera
        ubo = sr.scene.default_camera_ubo(W, H, eye=sr.scene.orbit_eye(rank, max(world, 1)) if world > 1 else (2.0, 2.0, 2.0))

    dev = sr.Device(0 if CPU_REHEARSAL else local_rank)
    mesh = sr.create_mesh_from_data(dev, verts, idx)
    tex = [sr.create_texture(dev, im, sampler=sr.default_sampler(args.max_anisotropy)) if im is not None else None for im in images]
    tgt = sr.RenderTarget(dev, W, H, samples=samples, n_views=views)

    def barrier():
        dev.synchronize()
        if world > 1:
            gpu_synchronize()
            dist.barrier()

    def timed(textures, steps):
        dev.synchronize()    # (the barrier across ranks, where there are several, is the caller's: timed_alone_then_together)
        t0 = time.perf_counter()
        for _ in range(steps):
            sr.draw(dev, tgt, mesh, ubo, textures)
        dev.synchronize()
        return time.perf_counter() - t0

    for _ in range(max(args.warmup, 1)):   # the first draw also sizes the tile reference list
        sr.draw(dev, tgt, mesh, ubo, tex)
    stats = tgt.stats()
    stats["samples"] = samples
    stats["views"] = views
    if world > 1:
        from softraster import distributed as d
        # rank 0 first runs its share by itself (the one-GPU point of this series, same workload), the others waiting at a barrier;
        # then every rank runs its share from a common barrier: max over the ranks
        alone_s, elapsed, own_elapsed = d.timed_alone_then_together(lambda: timed(tex, args.steps), rank, world, device=tensor_device(local_rank))
    else:
        elapsed = own_elapsed = alone_s = timed(tex, args.steps)
    if world > 1:
        gpu_synchronize()
        pv = torch.tensor([float(stats["shaded_pixels"])], dtype=torch.float64, device=tensor_device(local_rank))
        dist.all_reduce(pv, op=dist.ReduceOp.SUM)
        shaded_total = float(pv.item())
        every = [torch.zeros(1, dtype=torch.float64, device=tensor_device(local_rank)) for _ in range(world)]
        dist.all_gather(every, torch.tensor([own_elapsed], dtype=torch.float64, device=tensor_device(local_rank)))
        per_rank_s = [float(e.item()) for e in every]
    else:
        shaded_total = float(stats["shaded_pixels"])
        per_rank_s = [elapsed]

    # per-kernel durations from HIP events on the library's own stream, same K steps
    dev.profile(True)
    acc = {}
    for _ in range(args.steps):
        sr.draw(dev, tgt, mesh, ubo, tex)
        for k, v in dev.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    dev.profile(False)
    stage_ms = {k: v / args.steps for k, v in acc.items()}

    if rank == 0:
        T = stats["triangles"]; V = stats["vertices"]; R = stats["tile_refs"]; F = stats["visible_triangles"]
        P = stats["pixels"]; Pv = stats["shaded_pixels"]; surv = stats["triangles_binned"]
        ms_per_step = elapsed / args.steps * 1e3
        frames_per_s = world * views * args.steps / elapsed
        mtris = (T // views) * frames_per_s / 1e6
        tile = 32 if max(W, H) <= 8192 or samples != 1 else 64
        n_tiles = ((W + tile - 1) // tile) * ((H + tile - 1) // tile) * views
        by = kernel_bytes(stats, n_tiles)
        kernels = {k: stage_ms[k + "_ms"] for k in STAGES}
        if views <= 8 and V:     # the start of the pass rides along with the vertex kernel (a draw of a few views): one launch -- which the
            # library's events book under the first of the two stages --, both byte counts
            kernels["vertex"] += kernels["clear"]; kernels["clear"] = 0.0
            by["vertex"] += by["clear"]; by["clear"] = 0
        dom = max(kernels, key=kernels.get)
        traffic = measured_traffic(workload, dom)
        achieved = by[dom] / (kernels[dom] * 1e-3) / 1e9
        frame_bytes = frame_bytes_reference_model(stats)
        sampler = "trilinear sampler" if args.max_anisotropy <= 1.0 else f"maxAnisotropy {args.max_anisotropy:g} sampler"
        metric = {"c3_4k_10Mtri": f"Mtris/s, 3840x2160 frame of a 10M-triangle mesh with 4096^2 mip-mapped texture ({sampler})",
                  "c5_50Mtri_views": "Mtris/s, 1024-camera batch of a 50M-triangle mesh at 1920x1080, sharded by view"}.get(workload, "Mtris/s")
        out = {
            "metric": metric,
            "value": round(mtris, 2), "unit": "Mtris/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": workload, "width": W, "height": H, "triangles": T // views, "vertices": V // views, "samples": samples,
                       "views_per_gpu": views, "sharding": "by view (no collective)" if world > 1 else "single GPU",
                       "max_anisotropy": args.max_anisotropy},
            "frames_per_s": round(frames_per_s, 2),
            "gpixels_shaded_per_s": round(shaded_total * args.steps / elapsed / 1e9, 4),
            "per_gpu_Mtris_per_s": [round((T // views) * views * args.steps / s / 1e6, 1) for s in per_rank_s],
            "counters": {"V": V, "T": T, "R": R, "F": F, "X": int(Pv * 3 * 1.33), "P": P, "Pv": Pv, "S": samples, "binned": surv, "direct": stats["triangles_direct"],
                         "clipped": stats["triangles_clipped"],
                         "note": "X (unique texels over the three maps) is the model Pv*3*1.33, not counted; R = tile references, the direct path adds `direct`"},
            "kernel_ms": {k: round(v, 4) for k, v in kernels.items()},
            "kernel_algorithmic_MB": {k: round(v / 1e6, 1) for k, v in by.items()},
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_hbm_fraction": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            **({"frame_hbm_fraction_note": "the frame figure counts every view's mesh reads; a batch reads positions and indices once per pass of "
                                             f"{-(-views // max(stats.get('passes', 1), 1))} views, so this is not a fraction of what the kernels had to move"} if views > 1 else {}),
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes": int(by[dom]),
                         # what the kernel really moved through HBM (PMC counters) per second, as a fraction of the peak: NOT the same as
                         # `frac`, which prices the measurement plan's compulsory bytes -- the kernel skips the visibility words and the
                         # colour of blocks that hold nothing, so it moves fewer bytes than the plan counts
                         "hbm_utilisation": round(traffic / (kernels[dom] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if traffic else None,
                         "kernel_source_digest": kernel_source_digest()},
        }
        all_traffic = {k: measured_traffic(workload, k) for k in ("vertex", "setup", "fragment")}
        if all(all_traffic.values()):
            out["frame_hbm_utilisation"] = round(sum(all_traffic.values()) / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if world > 1:
            n1 = (T // views) * views * args.steps / alone_s / 1e6
            out["n1_same_workload_Mtris_per_s"] = round(n1, 1)
            out["n1_same_workload_ms_per_step"] = round(alone_s / args.steps * 1e3, 4)
            out["speedup_vs_n1_same_workload"] = round(mtris / n1, 3)
            out["n1_note"] = "rank 0 ran its own share alone, before the joint run, while the other ranks waited at a barrier (weak scaling: N ranks draw N shares)"
            # the default one-GPU line is the 4K headline frame, a different workload: the one-GPU point of THIS series is
            # `python bench.py --workload <this workload>`, and with no collective on the data path every rank's own rate
            # (per_gpu_Mtris_per_s) is what it measures
            out["scaling_note"] = (f"weak scaling by view, no collective: the N=1 point of this series is `python bench.py --workload {workload}` "
                                   "(each entry of per_gpu_Mtris_per_s is one GPU's rate on its share); the default N=1 line is the c3_4k_10Mtri headline, another workload")
        if world == 1 and views == 1 and not args.headline_only:
            # The same draw under a moving camera (the reference's own use: an orbit driven by the keys, src/main.cpp:223-267, the block
            # rewritten every frame, :988-1004): the eye advances one degree along the orbit every step, and, second figure, jumps a third
            # of the way round it every step.  The timed loop above redraws one camera; what it leans on from frame to frame (lean
            # draws, clears directed by the previous frame's flags, colour stores skipped where a block stays background) is counted here.
            def moving(eyes):
                ubos = [sr.scene.default_camera_ubo(W, H, eye=e) for e in eyes]
                for u in ubos[:max(args.warmup, 1)]:
                    sr.draw(dev, tgt, mesh, u, tex)
                dev.synchronize()
                before = dev.draw_counts()
                t0 = time.perf_counter()
                for u in ubos[max(args.warmup, 1):]:
                    sr.draw(dev, tgt, mesh, u, tex)
                dev.synchronize()
                dt = time.perf_counter() - t0
                after = dev.draw_counts()
                return dt / args.steps * 1e3, {k: after[k] - before[k] for k in after}
            n = args.steps + max(args.warmup, 1)
            orbit_ms, orbit_counts = moving([sr.scene.orbit_eye(k, 360) for k in range(n)])
            jump_ms, jump_counts = moving([sr.scene.orbit_eye((k * 127) % 360, 360) for k in range(n)])
            out["ms_per_step_orbit"] = round(orbit_ms, 4)
            out["ms_per_step_camera_jumps"] = round(jump_ms, 4)
            out["moving_camera"] = {"orbit": dict(orbit_counts, step_degrees=1), "jumps": dict(jump_counts, step_degrees=127),
                                    "note": "draws / lean_draws / lean_redraws / overflow_redraws of the timed steps; a lean redraw is a frame that had to be issued a second time in full"}
        if world == 1 and args.max_anisotropy <= 1.0 and not args.headline_only:
            # the same frame through the sampler the reference creates (maxAnisotropy 16): a second figure, not the headline
            tex16 = [sr.create_texture(dev, im, sampler=sr.default_sampler(16.0)) if im is not None else None for im in images]
            sr.draw(dev, tgt, mesh, ubo, tex16)
            out["ms_per_step_max_anisotropy_16"] = round(timed(tex16, args.steps) / args.steps * 1e3, 4)
            for t in tex16:
                if t is not None:
                    t.destroy()
        if not args.no_cpu_baseline and not args.headline_only and world == 1:   # reported at N=1 only: the other ranks would sit in the final barrier meanwhile
            from oracle import oracle
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            oracle.set_anisotropic(args.max_anisotropy > 1.0)
            chains = [(oracle.build_mip_chain(im), im.shape[1], im.shape[0]) if im is not None else (np.zeros((1, 4), np.uint8), 1, 1) for im in images]
            times = []
            for _ in range(3):      # (a frame is about 5 core-seconds on the 16 threads: three of them, the median reported)
                tc = time.perf_counter()
                oracle.draw(ubo if views == 1 else ubo[0], verts, idx, chains, W, H, samples, threads=threads)
                times.append(time.perf_counter() - tc)
            cpu_s = sorted(times)[1]
            out["cpu_baseline"] = {"value": round((T // views) / cpu_s / 1e6, 3), "unit": "Mtris/s", "cores": threads, "kind": "port",
                                   "sample": f"three full {W}x{H} frames of the same scene (mip chains prebuilt), median {cpu_s:.2f} s each",
                                   "what": "CPU restatement baseline (oracle/), not lavapipe", "lavapipe": lavapipe_status()}
        print(json.dumps(out), flush=True)

    barrier()
    for t in tex:
        if t is not None:
            t.destroy()
    tgt.destroy(); mesh.destroy(); dev.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
