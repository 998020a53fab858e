#!/usr/bin/env python3
"""Time from host buffers to the first finished frame of the headline workload (mesh and textures handed over in host memory:
uploads, position split, mip chains, first draw), next to the steady-state frame.  Printed as one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-renderer_amd")); sys.path.insert(0, ROOT)
import softraster as sr
W, H = 3840, 2160
verts, idx = sr.scene.uv_sphere(2500, 2000)
images = [sr.scene.albedo_pattern(4096), None, sr.scene.gray_map(2048, 2048, 0, 254), sr.scene.gray_map(2048, 2048, 72, 255)]
ubo = sr.scene.default_camera_ubo(W, H)
dev = sr.Device(0)
wv, wi = sr.scene.uv_sphere(8, 8)
warm = sr.create_mesh_from_data(dev, wv, wi); warm.destroy()      # context and module load are not part of the figure
res = []
for rep in range(3):
    t0 = time.perf_counter()
    mesh = sr.create_mesh_from_data(dev, verts, idx)
    t1 = time.perf_counter()
    tex = [sr.create_texture(dev, im, sampler=sr.default_sampler(1.0)) if im is not None else None for im in images]
    t2 = time.perf_counter()
    tgt = sr.RenderTarget(dev, W, H, samples=1, n_views=1)
    sr.draw(dev, tgt, mesh, ubo, tex); dev.synchronize()
    t3 = time.perf_counter()
    for _ in range(3):
        sr.draw(dev, tgt, mesh, ubo, tex)
    dev.synchronize()
    t4 = time.perf_counter()
    for _ in range(20):
        sr.draw(dev, tgt, mesh, ubo, tex)
    dev.synchronize()
    t5 = time.perf_counter()
    res.append({"mesh_upload_ms": (t1 - t0) * 1e3, "textures_ms": (t2 - t1) * 1e3, "target_and_first_draw_ms": (t3 - t2) * 1e3,
                "host_buffers_to_first_frame_ms": (t3 - t0) * 1e3, "steady_frame_ms": (t5 - t4) / 20 * 1e3})
    for t in tex:
        if t is not None:
            t.destroy()
    tgt.destroy(); mesh.destroy()
host_bytes = verts.nbytes + idx.nbytes + sum(im.nbytes for im in images if im is not None)
best = min(res, key=lambda r: r["host_buffers_to_first_frame_ms"])
T = idx.size // 3
print(json.dumps({"host_bytes": host_bytes, "runs": [{k: round(v, 3) for k, v in r.items()} for r in res],
                  "pcie_inclusive_Mtris_per_s": round(T / best["host_buffers_to_first_frame_ms"] / 1e3, 1),
                  "upload_GB_per_s": round(host_bytes / ((best["mesh_upload_ms"] + best["textures_ms"]) * 1e-3) / 1e9, 2)}))
dev.close()
