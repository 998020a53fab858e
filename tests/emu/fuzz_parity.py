"""TEST INFRASTRUCTURE ONLY: differential fuzzing of the kernels (executed on the CPU, tests/emu) against the
oracle.  Random triangle soups with mixed sizes, duplicates and degenerate triangles, random model
matrices and cameras, random frame sizes, sample counts, stored rectangles and view counts; visibility
must be bit-exact, colour within 1 LSB on 99.9 % of the pixels (on frames with enough covered pixels for
the percentage to mean something).

    python tests/emu/fuzz_parity.py [first_seed [count]]        SR_FUZZ_ANISOTROPIC=1: samplers with maxAnisotropy 16 on both sides
                                                                SR_FUZZ_GPU=1: the shipped library on GPU 0 instead of the CPU execution
                                                                of the kernel sources (same cases, same checks; a GPU box only)
                                                                SR_FUZZ_MANY_VIEWS=1: 9 to 14 views per case instead of 1 to 3 (with SR_FILL_BLOCKS=1 the
                                                                workgroups of the vertex and set-up kernels loop over them; with SR_TEST_DRAWS=2 the scenes of
                                                                small triangles are lean on the second draw and store no depth plane)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "vulkan-renderer_amd"), os.path.join(ROOT, "tests"), HERE]


def random_scene(rng):
    kind = rng.integers(0, 4)
    n = int(rng.integers(1, 400))
    if kind == 0:      # soup: mostly small, a few huge
        centers = rng.uniform(-1.5, 1.5, size=(n, 1, 3))
        scale = 2.5 * rng.uniform(0.0, 1.0, size=(n, 1, 1)) ** 4
        pos = centers + rng.uniform(-1, 1, size=(n, 3, 3)) * scale
    elif kind == 1:    # sub-pixel dust on a plane grid (ties, shared positions)
        grid = rng.integers(-40, 40, size=(n, 3, 3)) / rng.choice([16.0, 64.0, 257.0])
        pos = grid
    elif kind == 2:    # long slivers
        a = rng.uniform(-2, 2, size=(n, 1, 3))
        d = rng.normal(size=(n, 1, 3)) * rng.uniform(0.1, 3.0, size=(n, 1, 1))
        pos = np.concatenate([a, a + d, a + d * rng.uniform(0.2, 0.9, size=(n, 1, 1)) + rng.normal(size=(n, 1, 3)) * 1e-3], axis=1)
    else:              # a strip of connected triangles (shared edges)
        pts = np.cumsum(rng.normal(size=(n + 2, 3)) * 0.15, axis=0)
        pos = np.stack([pts[:-2], pts[1:-1], pts[2:]], axis=1)
        pos[1::2] = pos[1::2, ::-1]
    if rng.integers(0, 6) == 0:   # a few enormous triangles through the scene: vertices far outside the guard band, behind the eye, ...
        k = int(rng.integers(1, 4))
        pos = np.concatenate([pos, rng.uniform(-1.5, 1.5, size=(k, 1, 3)) + rng.uniform(-1, 1, size=(k, 3, 3)) * 10.0 ** rng.uniform(2, 4)])
        n += k
    pos = pos.astype(np.float32)
    dup = rng.integers(0, n, size=max(1, n // 9))
    pos[dup] = pos[(dup + 1) % n]                       # exact duplicates: depth ties
    deg = rng.integers(0, n, size=max(1, n // 11))
    pos[deg, 2] = pos[deg, 1]                           # degenerate triangles
    verts = np.zeros((n * 3, 8), np.float32)
    verts[:, 0:3] = pos.reshape(-1, 3)
    nrm = rng.normal(size=(n * 3, 3)); verts[:, 3:6] = (nrm / np.linalg.norm(nrm, axis=1, keepdims=True)).astype(np.float32)
    verts[:, 6:8] = rng.uniform(-3, 4, size=(n * 3, 2))
    idx = np.arange(n * 3, dtype=np.uint32)
    if rng.integers(0, 2):   # share vertices through the index list
        idx = rng.permutation(idx)[: (n * 3) // 3 * 3].astype(np.uint32) if rng.integers(0, 2) else np.minimum(idx, rng.integers(3, n * 3 + 1) - 1).astype(np.uint32)
    return verts, idx


def scene_of_seed(seed, sr):
    """Everything one fuzz case consists of, as a dict; deterministic in the seed."""
    import test_gpu_parity as T
    rng = np.random.default_rng(seed)
    verts, idx = random_scene(rng)
    W, H = int(rng.integers(1, 300)), int(rng.integers(1, 200))
    if seed % 16 == 5:   # now and then a frame of many tiles
        W, H = int(rng.integers(600, 1500)), int(rng.integers(400, 900))
    samples = int(rng.choice([1, 4]))
    n_views = int(rng.integers(1, 4))
    if os.environ.get("SR_FUZZ_MANY_VIEWS") == "1":   # batches of more views than travel as kernel arguments (the same scenes otherwise: the draw follows the others)
        n_views = int(np.random.default_rng(seed + 7919).integers(9, 15))
    rect = None
    if rng.integers(0, 3) == 0:
        x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
        rect = (x, y, int(rng.integers(1, W - x + 1)), int(rng.integers(1, H - y + 1)))
    ubos = np.stack([T._random_view(rng, W, H, sr) for _ in range(n_views)])
    if rng.integers(0, 4) == 0:   # a camera inside the cloud: vertices behind the eye
        ubos[0] = sr.scene.default_camera_ubo(W, H, eye=(0.1, 0.05, 0.02))
    size = [int(rng.choice([1, 2, 3, 8, 17, 64])) for _ in range(2)]
    images = [sr.scene.noise_rgba(size[0], size[1], seed), sr.scene.constant_rgba(1, 1, (128, 128, 255, 255)),
              sr.scene.noise_rgba(size[1], size[0], seed + 1), sr.scene.noise_rgba(size[0], size[0], seed + 2)]
    return {"verts": verts, "idx": idx, "W": W, "H": H, "samples": samples, "rect": rect, "ubos": ubos, "images": images}


def check_seed(seed, sr, oracle, dev, util, vertex_stage_too=False):
    """Draws the case of `seed` through the C ABI and with the oracle: visibility and the clipped / given-up counts must
    be equal, colour must meet the bar on every frame with enough covered pixels for a percentage to mean something.
    Returns (lowest colour fraction over such frames, triangles clipped, triangles given up)."""
    sc = scene_of_seed(seed, sr)
    verts, idx, W, H, samples, rect, ubos, images = (sc[k] for k in ("verts", "idx", "W", "H", "samples", "rect", "ubos", "images"))
    if vertex_stage_too:   # the vertex stage on its own: clip coordinates and window-space records, bit for bit
        mesh = sr.create_mesh_from_data(dev, verts, idx)
        clip, screen = mesh.vertex_stage(ubos[0], W, H)
        rclip, rscreen = oracle.vertex_stage(ubos[0], verts, W, H)
        mesh.destroy()
        assert np.array_equal(clip.view(np.uint32), rclip.view(np.uint32)), f"seed {seed}: clip coordinates differ"
        for field in ("X", "Y", "z", "rw"):
            assert np.array_equal(screen[field].view(np.uint32), rscreen[field].view(np.uint32)), f"seed {seed}: window-space {field} differs"
    got = util.gpu_render(sr, dev, verts, idx, images, ubos, W, H, samples, rect=rect)
    ref_dropped = ref_clipped = 0
    lowest = 1.0
    for v in range(len(ubos)):
        ref = oracle.draw(ubos[v], verts, idx, images, W, H, samples, rect=rect)
        what = f"seed {seed} view {v}: {idx.size // 3} triangles, {W}x{H}x{samples}, rect {rect}"
        assert np.array_equal(got[v]["tri"], ref["tri"]), what + f": triangle ids differ at {int((got[v]['tri'] != ref['tri']).sum())} samples"
        assert np.array_equal(got[v]["depth"].view(np.uint32), ref["depth"].view(np.uint32)), what + ": depth bits differ"
        ref_dropped += ref["stats"]["tris_clip_culled"]; ref_clipped += ref["stats"]["tris_clipped"]
        covered = int((ref["tri"] != 0xFFFFFFFF).any(axis=-1).sum())
        frac, worst = util.color_match_fraction(got[v]["bgra"], ref["bgra"], 1)
        if covered >= 2000:
            lowest = min(lowest, frac)
            assert frac >= 0.999, what + f": colour within 1 LSB on {frac:.5f} (worst {worst})"
    st = got[0]["stats"]
    # dropped are only the triangles the near-plane clipper gives up on
    assert st["triangles_unprojectable"] == ref_dropped and st["triangles_clipped"] == ref_clipped, f"seed {seed}: dropped / clipped counts {st} vs {ref_dropped}, {ref_clipped}"
    return lowest, ref_clipped, ref_dropped


def main():
    import softraster as sr
    import util
    from oracle import oracle
    on_gpu = os.environ.get("SR_FUZZ_GPU") == "1"
    if not on_gpu:
        import build_emu
        sr.LIB_PATH = build_emu.build()
    aniso = os.environ.get("SR_FUZZ_ANISOTROPIC") == "1"
    sr.DEFAULT_MAX_ANISOTROPY = 16.0 if aniso else 1.0
    oracle.set_anisotropic(aniso)
    oracle.set_near_clip(True)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    dev = sr.Device(0)
    worst_frac = 1.0
    total_clipped = total_dropped = 0
    for seed in range(first, first + count):
        lowest, clipped, dropped = check_seed(seed, sr, oracle, dev, util, vertex_stage_too=seed % 4 == 0)
        worst_frac = min(worst_frac, lowest)
        total_clipped += clipped; total_dropped += dropped
        if on_gpu and (seed - first) % 50 == 49:
            print(f"... through seed {seed}, lowest colour fraction so far {worst_frac:.5f}", flush=True)
    print(f"ok: seeds {first}..{first + count - 1} on {'the GPU' if on_gpu else 'the CPU execution of the kernels'}, lowest colour fraction on well-covered frames {worst_frac:.5f}"
          f"; near-plane clipper: {total_clipped} triangles clipped, {total_dropped} given up")


if __name__ == "__main__":
    main()
