"""Helpers shared by the parity tests."""
import numpy as np

GOLDEN_DIR = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.abspath(__file__)), "golden")


def random_soup(n_tris, seed, lo=-1.2, hi=1.2, size=0.25, z_spread=1.0):
    """Random triangle soup around the origin: independent vertices per triangle, mixed sizes and
    orientations, with deliberate exact duplicates (coplanar ties) every 7th triangle."""
    rng = np.random.default_rng(seed)
    centers = rng.uniform(lo, hi, size=(n_tris, 1, 3)).astype(np.float32)
    centers[..., 2] *= z_spread
    scale = (size * rng.uniform(0.02, 1.0, size=(n_tris, 1, 1)) ** 2).astype(np.float32)
    offs = rng.uniform(-1, 1, size=(n_tris, 3, 3)).astype(np.float32) * scale
    pos = centers + offs
    dup = np.arange(7, n_tris, 7)
    pos[dup] = pos[dup - 1]
    verts = np.zeros((n_tris * 3, 8), np.float32)
    verts[:, 0:3] = pos.reshape(-1, 3)
    nrm = rng.normal(size=(n_tris * 3, 3)).astype(np.float32)
    verts[:, 3:6] = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    verts[:, 6:8] = rng.uniform(-2, 3, size=(n_tris * 3, 2)).astype(np.float32)
    idx = np.arange(n_tris * 3, dtype=np.uint32)
    return verts, idx


def color_match_fraction(a, b, tol=1):
    d = np.abs(a.astype(np.int32) - b.astype(np.int32)).max(axis=-1)
    return float((d <= tol).mean()), int(d.max())


def gpu_render(sr, device, verts, idx, images, ubos, width, height, samples=1, rect=None, textures=None, draws=None):
    """Draw through the C ABI and read everything back.  Returns a list of per-view dicts.  draws (default: SR_TEST_DRAWS or 1):
    how often the frame is drawn into the target before it is read -- a second draw takes the paths that go by the first one's
    counters (lean draws, the direct rasteriser)."""
    import os
    draws = draws or int(os.environ.get("SR_TEST_DRAWS", "1"))
    ubos = np.atleast_2d(np.asarray(ubos, np.float32))
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = textures if textures is not None else [sr.create_texture(device, im) for im in images]
    tgt = sr.RenderTarget(device, width, height, samples=samples, n_views=ubos.shape[0], rect=rect)
    try:
        for _ in range(draws):
            sr.draw(device, tgt, mesh, ubos, tex)
        out = [{"tri": tgt.read_triangle_ids(v), "depth": tgt.read_depth(v), "bgra": tgt.read_color(v), "vis": tgt.read_visibility(v)}
               for v in range(ubos.shape[0])]
        stats = tgt.stats()
        for o in out:
            o["stats"] = stats
        return out
    finally:
        tgt.destroy()
        mesh.destroy()
        if textures is None:
            for t in tex:
                t.destroy()


def assert_visibility_equal(got, ref):
    assert np.array_equal(got["tri"], ref["tri"]), f"triangle ids differ at {int((got['tri'] != ref['tri']).sum())} samples"
    assert np.array_equal(got["depth"].view(np.uint32), ref["depth"].view(np.uint32)), \
        f"depth bits differ at {int((got['depth'].view(np.uint32) != ref['depth'].view(np.uint32)).sum())} samples"


def device_with_env(sr, **env):
    """A device handle created while the given environment variables are set (sr_device_create reads its settings
    from the environment); the environment is put back as it was."""
    import os
    saved = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        return sr.Device(0)
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
