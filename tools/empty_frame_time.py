import sys, os, numpy as np
sys.path.insert(0, "vulkan-renderer_amd"); sys.path.insert(0, ".")
import softraster as sr
W, H = 3840, 2160
verts, idx = sr.scene.uv_sphere(250, 200)
images = [sr.scene.albedo_pattern(1024), None, sr.scene.gray_map(512, 512, 0, 254), sr.scene.gray_map(512, 512, 72, 255)]
dev = sr.Device(0)
mesh = sr.create_mesh_from_data(dev, verts, idx)
tex = [sr.create_texture(dev, im, sampler=sr.default_sampler(1.0)) if im is not None else None for im in images]
for (w, h) in ((3840, 2160), (7680, 4320), (1920, 1080)):
    tgt = sr.RenderTarget(dev, w, h)
    model = np.eye(4, dtype=np.float32); model[0, 3] = 100.0
    for name, ubo in (("empty", sr.scene.default_camera_ubo(w, h, model=model)), ("sphere", sr.scene.default_camera_ubo(w, h))):
        for _ in range(5): sr.draw(dev, tgt, mesh, ubo, tex)
        dev.profile(True); acc = 0.0
        for _ in range(50):
            sr.draw(dev, tgt, mesh, ubo, tex); acc += dev.stage_times()["fragment_ms"]
        dev.profile(False)
        st = tgt.stats()
        print(w, h, name, "fragment_ms %.4f" % (acc / 50), "shaded", st["shaded_pixels"], "blocks", ((w + 31) // 32) * ((h + 7) // 8))
    tgt.destroy()
