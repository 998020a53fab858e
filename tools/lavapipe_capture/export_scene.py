#!/usr/bin/env python3
"""Writes a scene directory for tools/lavapipe_capture/capture: the synthetic scenes the parity tests use (UV sphere,
procedural textures, the reference's default camera), as plain binary files.

    python tools/lavapipe_capture/export_scene.py OUT_DIR [--width 800 --height 600 --segments 88 --samples 1
                                                           --max-anisotropy 16 --texture 256]

    OUT_DIR/scene.txt      width, height, samples, max_anisotropy, n_vertices, n_indices, tex0..tex3 sizes
    OUT_DIR/vertices.bin   32-byte records (position, normal, uv: src/mesh.hpp:18-54 of the reference)
    OUT_DIR/indices.bin    uint32 triangle list
    OUT_DIR/ubo.bin        the 208-byte uniform block (src/main.cpp:50-55, 988-1004 of the reference)
    OUT_DIR/tex{0..3}.rgba RGBA8 albedo, normal, metallic, roughness

tests/test_lavapipe_captures.py reads the same directory back, so a capture is compared on exactly these inputs."""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "vulkan-renderer_amd"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("out")
    ap.add_argument("--width", type=int, default=800)
    ap.add_argument("--height", type=int, default=600)
    ap.add_argument("--segments", type=int, default=88)
    ap.add_argument("--samples", type=int, default=1, choices=[1, 4])
    ap.add_argument("--max-anisotropy", type=float, default=16.0)
    ap.add_argument("--texture", type=int, default=256)
    args = ap.parse_args()
    from softraster import scene     # numpy only: no GPU library is loaded
    verts, idx = scene.uv_sphere(args.segments, args.segments)
    ubo = scene.default_camera_ubo(args.width, args.height)
    images = scene.material_textures(args.texture)
    if images[1] is None:
        images[1] = scene.constant_rgba(1, 1, (128, 128, 255, 255))   # the normal map is bound but never sampled
    os.makedirs(args.out, exist_ok=True)
    np.ascontiguousarray(verts, np.float32).tofile(os.path.join(args.out, "vertices.bin"))
    np.ascontiguousarray(idx, np.uint32).tofile(os.path.join(args.out, "indices.bin"))
    block = np.zeros(52, np.float32)
    flat = np.ascontiguousarray(ubo, np.float32).reshape(-1)
    block[:flat.size] = flat
    block.tofile(os.path.join(args.out, "ubo.bin"))
    with open(os.path.join(args.out, "scene.txt"), "w") as f:
        f.write(f"width {args.width}\nheight {args.height}\nsamples {args.samples}\nmax_anisotropy {args.max_anisotropy}\n")
        f.write(f"n_vertices {verts.shape[0]}\nn_indices {idx.size}\n")
        for i, im in enumerate(images):
            im = np.ascontiguousarray(im, np.uint8)
            im.tofile(os.path.join(args.out, f"tex{i}.rgba"))
            f.write(f"tex{i} {im.shape[1]} {im.shape[0]}\n")
    print(f"{args.out}: {verts.shape[0]} vertices, {idx.size // 3} triangles, {args.width}x{args.height}x{args.samples}")


if __name__ == "__main__":
    main()
