#!/usr/bin/env python3
"""Benchmark of the draw path: BASELINE config 3 -- one 3840x2160 frame of a 10 M-triangle UV sphere
with a 4096^2 mip-mapped albedo map (2048^2 metallic / roughness), default reference camera.

    python bench.py --gpus N --steps K --warmup W

A step is one sr_draw (clear, vertex, set-up + binning, raster, fragment) of one view per GPU with
mesh and textures already resident in HBM.  With N > 1 (launched by torch.distributed.run, one rank
per GPU) every rank draws its own camera of the same orbit: the path shards by view with no
collective on the data path (weak scaling).  Rank 0 prints one JSON line.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "vulkan-renderer_amd"))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak

WORKLOADS = {
    # name: (width, height, sphere X, sphere Y, albedo size, other maps size, samples, views per GPU)
    "c3_4k_10Mtri": (3840, 2160, 2500, 2000, 4096, 2048, 1, 1),          # the headline: BASELINE config 3
    "small_1080p_124ktri": (1920, 1080, 250, 248, 1024, 512, 1, 1),       # for quick checks only
    "c1_800x600_msaa4": (800, 600, 88, 88, 2048, 2048, 4, 1),             # the reference's own frame: large triangles, 4 samples
    "c2_1080p_64views": (1920, 1080, 88, 88, 2048, 2048, 1, 64),          # BASELINE config 2: one draw of 64 cameras
}


def kernel_bytes(st, n_tiles):
    """Algorithmic (compulsory) HBM bytes of each kernel for one frame: every array counted once per
    kernel that must touch it, caches assumed perfect inside a kernel.  st: counters of the draw."""
    V, T, P, S = st["vertices"], st["triangles"], st["pixels"], st.get("samples", 1)
    B, R, Fd, F, Pv = st["triangles_binned"], st["tile_refs"], st["triangles_direct"], st["visible_triangles"], st["shaded_pixels"]
    X = Pv * 3 * 1.33                       # unique texels over the three maps (model)
    direct_samples = Pv * S * (Fd / max(F, 1)) if F else 0
    return {
        "clear": P * S * 8,                                                   # visibility words written
        "vertex": V * 12 + V * 16,                                             # positions read, xy + (z, 1/w) planes written
        "setup": T * 12 + V * 8 + min(V, 3 * Fd) * 8 + direct_samples * 8 + B * 4,   # indices, xy plane, z of covered micro-triangles, atomic-min words, pending ids
        "bin": B * 4 + B * 12 + min(V, 3 * B) * 8 + R * 12,                   # pending ids, indices, xy, records
        "scan": n_tiles * 12,
        "fill": R * 12 + R * 4,                                                # records read, references written
        "raster": R * 4 + R * 12 + min(V, 3 * R) * 12 + (Pv * S - direct_samples) * 8,
        "fragment": P * S * 8 + F * 12 + min(V, 3 * F) * 48 + X * 4 + P * 4,  # visibility, indices, vertex records + window planes, texels, colour
    }


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


def frame_bytes_reference_model(st):
    """The per-frame figure of the measurement plan this benchmark follows: vertex V*28; set-up/bin
    T*12 + V*16 + R*4; raster R*4 + R*12 + V*16 + P*S*8; fragment P*S*8 + F*108 + X*4 + P*4, with R =
    surviving (triangle, tile) references (here: tile references + directly resolved triangles)."""
    V, T, P, S = st["vertices"], st["triangles"], st["pixels"], st.get("samples", 1)
    R = st["tile_refs"] + st["triangles_direct"]
    F, Pv = st["visible_triangles"], st["shaded_pixels"]
    X = Pv * 3 * 1.33
    return (V * 28) + (T * 12 + V * 16 + R * 4) + (R * 4 + R * 12 + V * 16 + P * S * 8) + (P * S * 8 + F * 108 + X * 4 + P * 4)


def sort_first(args, sr, rank, local_rank, world):
    """BASELINE config 4: the 10 M-triangle scene at 7680x4320, the frame split into horizontal bands over
    the ranks (mesh and textures replicated), finished BGRA8 bands gathered on rank 0 over RCCL."""
    import torch
    import torch.distributed as dist
    from softraster import distributed as d
    W, H = 7680, 4320
    verts, idx = sr.scene.uv_sphere(2500, 2000)
    images = [sr.scene.albedo_pattern(4096), None, sr.scene.gray_map(2048, 2048, 0, 254), sr.scene.gray_map(2048, 2048, 72, 255)]
    ubo = sr.scene.default_camera_ubo(W, H)
    torch.cuda.set_device(local_rank)
    dev = sr.Device(local_rank)
    mesh = sr.create_mesh_from_data(dev, verts, idx)
    tex = [sr.create_texture(dev, im) if im is not None else None for im in images]
    bands = d.split_bands(W, H, world)
    band = bands[rank]
    out = torch.empty((band[3], band[2], 4), dtype=torch.uint8, device=f"cuda:{local_rank}")
    tgt = sr.RenderTarget(dev, W, H, samples=1, n_views=1, rect=band)
    tgt.bind_color(out.data_ptr())

    def step():
        sr.draw(dev, tgt, mesh, ubo, tex)
        dev.synchronize()
        return d.gather_bands(out, bands, rank, dst=0) if world > 1 else out

    for _ in range(max(args.warmup, 1)):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        T = idx.size // 3
        print(json.dumps({
            "metric": "Mtris/s, 7680x4320 frame of a 10M-triangle mesh split sort-first over the GPUs", "value": round(T * args.steps / elapsed / 1e6, 2),
            "unit": "Mtris/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "c4_8k_sort_first", "width": W, "height": H, "triangles": T, "bands": bands,
                       "composite": "torch.distributed gather of BGRA8 bands (RCCL)" if world > 1 else "none"},
            "frames_per_s": round(args.steps / elapsed, 2)}), flush=True)
    for t_ in tex:
        if t_ is not None:
            t_.destroy()
    tgt.destroy(); mesh.destroy(); dev.close()
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3_4k_10Mtri", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="views", choices=["views", "sortfirst"],
                    help="views: every rank draws its own camera of the workload (weak scaling, no collective); "
                         "sortfirst: one 7680x4320 frame split into bands over the ranks, bands gathered on rank 0 (strong scaling)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")

    if world > 1:
        import torch
        import torch.distributed as dist
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import softraster as sr
    if args.mode == "sortfirst":
        return sort_first(args, sr, rank, local_rank, world)
    W, H, SX, SY, alb_size, map_size, samples, views = WORKLOADS[args.workload]
    verts, idx = sr.scene.uv_sphere(SX, SY)
    images = [sr.scene.albedo_pattern(alb_size), None, sr.scene.gray_map(map_size, map_size, 0, 254), sr.scene.gray_map(map_size, map_size, 72, 255)]
    # rank r draws view r of a `world`-camera orbit; view 0 is the reference's default camera
    ubo = sr.scene.default_camera_ubo(W, H, eye=sr.scene.orbit_eye(rank, max(world, 1)) if world > 1 else (2.0, 2.0, 2.0))
    if views > 1:   # a batch: this rank's share of a world*views-camera orbit
        ubo = sr.scene.orbit_ubos(views, W, H, first=rank * views, total=world * views)

    dev = sr.Device(local_rank)
    mesh = sr.create_mesh_from_data(dev, verts, idx)
    tex = [sr.create_texture(dev, im) if im is not None else None for im in images]
    tgt = sr.RenderTarget(dev, W, H, samples=samples, n_views=views)

    def barrier():
        dev.synchronize()
        if world > 1:
            torch.cuda.synchronize()
            dist.barrier()

    for _ in range(max(args.warmup, 1)):   # the first draw also sizes the tile reference list
        sr.draw(dev, tgt, mesh, ubo, tex)
    stats = tgt.stats()
    stats["samples"] = samples
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sr.draw(dev, tgt, mesh, ubo, tex)
    dev.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        torch.cuda.synchronize()
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        pv = torch.tensor([float(stats["shaded_pixels"])], dtype=torch.float64, device="cuda")
        dist.all_reduce(pv, op=dist.ReduceOp.SUM)
        shaded_total = float(pv.item())
    else:
        shaded_total = float(stats["shaded_pixels"])

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
        tile = 64 if samples == 1 else 32
        n_tiles = ((W + tile - 1) // tile) * ((H + tile - 1) // tile) * views
        by = kernel_bytes(stats, n_tiles)
        kernels = {k: stage_ms[k + "_ms"] for k in ("clear", "vertex", "setup", "bin", "scan", "fill", "raster", "fragment")}
        dom = max(kernels, key=kernels.get)
        achieved = by[dom] / (kernels[dom] * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # HBM bytes per launch from rocprofv3 PMC passes, if collected
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(args.workload, {}).get(dom)
            except Exception:
                traffic = None
        frame_bytes = frame_bytes_reference_model(stats)
        out = {
            "metric": "Mtris/s, 3840x2160 frame of a 10M-triangle mesh with 4096^2 mip-mapped texture" if args.workload == "c3_4k_10Mtri" else "Mtris/s",
            "value": round(mtris, 2), "unit": "Mtris/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": args.workload, "width": W, "height": H, "triangles": T // views, "vertices": V // views, "samples": samples,
                       "views_per_gpu": views, "sharding": "by view (no collective)" if world > 1 else "single GPU",
                       "pipeline_variant": os.environ.get("SR_VARIANT", "") or "default"},
            "frames_per_s": round(frames_per_s, 2),
            "gpixels_shaded_per_s": round(shaded_total * args.steps / elapsed / 1e9, 4),
            "counters": {"V": V, "T": T, "R": R, "F": F, "X": int(Pv * 3 * 1.33), "P": P, "Pv": Pv, "S": samples, "binned": surv, "direct": stats["triangles_direct"],
                         "note": "X (unique texels over the three maps) is the model Pv*3*1.33, not counted; R = tile references, the direct path adds `direct`"},
            "kernel_ms": {k: round(v, 4) for k, v in kernels.items()},
            "kernel_algorithmic_MB": {k: round(v / 1e6, 1) for k, v in by.items()},
            "frame_algorithmic_bytes": int(frame_bytes),
            "frame_hbm_fraction": round(frame_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "roofline": {"kernel": dom, "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "algorithmic_bytes": int(by[dom])},
        }
        if not args.no_cpu_baseline and world == 1:   # reported at N=1 only: the other ranks would sit in the final barrier meanwhile
            from oracle import oracle
            threads = args.cpu_threads or min(16, os.cpu_count() or 1)
            chains = [(oracle.build_mip_chain(im), im.shape[1], im.shape[0]) if im is not None else (np.zeros((1, 4), np.uint8), 1, 1) for im in images]
            tc = time.perf_counter()
            oracle.draw(ubo if views == 1 else ubo[0], verts, idx, chains, W, H, samples, threads=threads)
            cpu_s = time.perf_counter() - tc
            out["cpu_baseline"] = {"value": round((T // views) / cpu_s / 1e6, 3), "unit": "Mtris/s", "cores": threads, "kind": "port",
                                   "sample": f"one full {W}x{H} frame of the same scene (mip chains prebuilt), {cpu_s:.2f} s",
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
