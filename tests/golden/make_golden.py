"""Generates the golden fixtures in this directory from the CPU oracle (oracle/sr_oracle.c).

The reference has no test vectors for the draw path and cannot be built or run where this was
written, so these vectors pin the oracle against ITSELF over time (and the GPU path against the
oracle); they do not pin either against the reference ("parity unpinned").  Inputs are procedural
and rebuilt by build_scene(); only the expected outputs are stored.

    python tests/golden/make_golden.py        # rewrites the .npz files
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, "vulkan-renderer_amd"), os.path.dirname(HERE)):
    if p not in sys.path:
        sys.path.insert(0, p)

SCENES = ("sphere_s1", "sphere_s4", "soup_s1", "soup_s4", "quad_checker")


def path_of(name):
    return os.path.join(HERE, name + ".npz")


def build_scene(name):
    from softraster import scene
    import util
    if name.startswith("sphere"):
        verts, idx = scene.uv_sphere(24, 16)
        return dict(verts=verts, idx=idx, images=scene.material_textures(32), ubo=scene.default_camera_ubo(96, 64),
                    width=96, height=64, samples=4 if name.endswith("s4") else 1)
    if name.startswith("soup"):
        verts, idx = util.random_soup(300, seed=1, size=0.9)
        return dict(verts=verts, idx=idx, images=scene.material_textures(16, seed=9), ubo=scene.default_camera_ubo(80, 60),
                    width=80, height=60, samples=4 if name.endswith("s4") else 1)
    if name == "quad_checker":
        verts, idx = scene.screen_quad(z=0.0, half=1.0, uv_scale=4.0)
        checker = np.zeros((2, 2, 4), np.uint8)
        checker[..., 3] = 255
        checker[0, 0, :3] = (255, 32, 0)
        checker[1, 1, :3] = (0, 200, 255)
        checker[0, 1, :3] = (90, 90, 90)
        gray = scene.gray_map(8, 8, 72, 255)
        return dict(verts=verts, idx=idx, images=[checker, scene.constant_rgba(2, 2, (128, 128, 255, 255)), scene.gray_map(4, 4, 0, 254), gray],
                    ubo=scene.default_camera_ubo(64, 64, eye=(0.4, -0.3, 3.0), up=(0.0, 1.0, 0.0)), width=64, height=64, samples=1)
    raise KeyError(name)


def main():
    from oracle import oracle
    for name in SCENES:
        sc = build_scene(name)
        r = oracle.draw(sc["ubo"], sc["verts"], sc["idx"], sc["images"], sc["width"], sc["height"], sc["samples"])
        np.savez_compressed(path_of(name), tri=r["tri"], depth=r["depth"], bgra=r["bgra"])
        print(name, r["stats"], os.path.getsize(path_of(name)), "bytes")


if __name__ == "__main__":
    main()
