#!/usr/bin/env python3
"""Sort-first rehearsal on ONE GPU: the 7680x4320 frame of the 10 M-triangle scene cut into the cost-balanced bands that 2, 4 and 8
ranks would draw (softraster.distributed.balance_bands), every band drawn and timed on this GPU by itself.  The slowest band of a
split is what a rank of that many GPUs would take per frame (every rank holds the whole mesh; the exchange of 16.6 MB bands is
overlapped with the next frame).  A projection from measured per-band times, not a multi-GPU measurement.  One JSON line.
Every band also reports how many of the mesh's chunks of 768 triangles the draw culled against the band (sr_draw_counts).
    --lib PATH      another build of the library          --counts-only   one draw per band, no timing (counters only)"""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vulkan-renderer_amd")); sys.path.insert(0, ROOT)
import softraster as sr
from softraster import distributed as d
if "--lib" in sys.argv:
    sr.LIB_PATH = os.path.abspath(sys.argv[sys.argv.index("--lib") + 1])
COUNTS_ONLY = "--counts-only" in sys.argv
W, H = 7680, 4320
verts, idx = sr.scene.uv_sphere(2500, 2000)
images = [sr.scene.albedo_pattern(4096), None, sr.scene.gray_map(2048, 2048, 0, 254), sr.scene.gray_map(2048, 2048, 72, 255)]
ubo = sr.scene.default_camera_ubo(W, H)
dev = sr.Device(0)
mesh = sr.create_mesh_from_data(dev, verts, idx)
tex = [sr.create_texture(dev, im, sampler=sr.default_sampler(1.0)) if im is not None else None for im in images]

def timed(rect, steps=20):
    tgt = sr.RenderTarget(dev, W, H, samples=1, n_views=1, rect=rect)
    before = dev.draw_counts()
    sr.draw(dev, tgt, mesh, ubo, tex)
    after = dev.draw_counts()
    cull = {"chunks_tested": after["chunks_tested"] - before["chunks_tested"], "chunks_culled": after["chunks_culled"] - before["chunks_culled"]}
    ms, st = 0.0, {}
    if not COUNTS_ONLY:
        for _ in range(2):
            sr.draw(dev, tgt, mesh, ubo, tex)
        dev.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            sr.draw(dev, tgt, mesh, ubo, tex)
        dev.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        dev.profile(True); sr.draw(dev, tgt, mesh, ubo, tex); st = dev.stage_times(); dev.profile(False)
    tri = tgt.read_triangle_ids(0)[..., 0]
    tgt.destroy()
    return ms, dict({k: round(v, 4) for k, v in st.items()}, **cull), (tri != sr.NO_TRIANGLE).sum(axis=1).astype(np.float64)

whole_ms, whole_stages, rows = timed(None)
out = {"frame": [W, H], "triangles": int(idx.size // 3), "one_gpu_ms": round(whole_ms, 4), "one_gpu_stage_ms": whole_stages, "splits": {}}
ROW_CONSTANT = float(os.environ.get("C4_ROW_CONSTANT", W / 32))   # cost of a row that holds nothing, in covered pixels
out["row_constant"] = ROW_CONSTANT
for n in (2, 4, 8):
    bands = d.balance_bands(ROW_CONSTANT + rows, W, n)
    per_band = []
    for b in bands:
        ms, stages, _ = timed(b)
        per_band.append({"band": list(b), "ms": round(ms, 4), "vertex_ms": stages.get("vertex_ms"), "clear_ms": stages.get("clear_ms"),
                         "setup_ms": stages.get("setup_ms"), "fragment_ms": stages.get("fragment_ms"),
                         "chunks_tested": stages["chunks_tested"], "chunks_culled": stages["chunks_culled"]})
    slowest = max(p["ms"] for p in per_band)
    out["splits"][str(n)] = {"bands": per_band, "slowest_band_ms": slowest, "projected_speedup": round(whole_ms / slowest, 2) if slowest else None}
print(json.dumps(out))
for t in tex:
    if t is not None:
        t.destroy()
mesh.destroy(); dev.close()
