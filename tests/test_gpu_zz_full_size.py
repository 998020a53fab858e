"""Parity at the size the headline number is quoted on: the 10 M-triangle sphere at 3840x2160 with a
4096^2 albedo map.  The CPU oracle cannot render that frame in test time, so this file checks
  * windows of the frame against the oracle (the oracle rasterises a sub-rectangle of the same 10 M
    triangles), bit-exact visibility and <= 1 LSB colour,
  * and size-independent properties of the whole frame: repeatability, independence of the screen
    split, independence of the triangle order, the closed-form silhouette of the sphere, and the
    work counters against the frame itself.
Collected last on purpose (file name): a failure here hides no other test.

Under the CPU execution of the kernels (SR_TEST_EMULATED=1) the same tests run on a scaled-down scene,
which checks the test logic itself without a GPU.
"""
import math
import os

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

SMALL = os.environ.get("SR_TEST_EMULATED") == "1" and os.environ.get("SR_EMULATE_FULL_SIZE") != "1"
#        W     H     X     Y     albedo  maps
CONFIG = (480, 270, 150, 120, 256, 128) if SMALL else (3840, 2160, 2500, 2000, 4096, 2048)


@pytest.fixture(scope="module")
def frame(sr, device):
    W, H, SX, SY, alb, maps = CONFIG
    verts, idx = sr.scene.uv_sphere(SX, SY)
    images = [sr.scene.albedo_pattern(alb), None, sr.scene.gray_map(maps, maps, 0, 254), sr.scene.gray_map(maps, maps, 72, 255)]
    tex = [sr.create_texture(device, im) if im is not None else None for im in images]
    mesh = sr.create_mesh_from_data(device, verts, idx)
    ubo = sr.scene.default_camera_ubo(W, H)
    tgt = sr.RenderTarget(device, W, H, samples=1, n_views=1)
    sr.draw(device, tgt, mesh, ubo, tex)
    f = {"W": W, "H": H, "verts": verts, "idx": idx, "images": images, "tex": tex, "mesh": mesh, "ubo": ubo, "tgt": tgt,
         "tri": tgt.read_triangle_ids(0)[..., 0], "depth": tgt.read_depth(0)[..., 0], "bgra": tgt.read_color(0), "stats": tgt.stats()}
    yield f
    tgt.destroy()
    mesh.destroy()
    for t in tex:
        if t is not None:
            t.destroy()


def test_counters_describe_the_frame(sr, frame):
    W, H = frame["W"], frame["H"]
    st, tri = frame["stats"], frame["tri"]
    T = frame["idx"].size // 3
    covered = tri != sr.NO_TRIANGLE
    assert st["triangles"] == T and st["vertices"] == frame["verts"].shape[0] and st["pixels"] == W * H
    assert st["shaded_pixels"] == int(covered.sum())
    assert st["visible_triangles"] == int(np.unique(tri[covered]).size)
    assert st["triangles_unprojectable"] == 0 and st["cull_violations"] == 0
    # about half of the sphere faces the camera; every surviving triangle went one of the two ways
    assert 0 < st["triangles_direct"] + st["triangles_binned"] <= T
    assert st["tile_refs"] >= st["triangles_binned"] or st["tile_refs"] == 0   # 0: resolved without tile lists (sparse-tail variant)
    assert int(tri[covered].max()) < T


def test_silhouette_is_the_projected_sphere(sr, frame):
    """A unit sphere seen from distance d through a 45-degree lens is a disc of radius
    (H/2) * tan(asin(1/d)) / tan(22.5 deg) pixels around the frame centre."""
    W, H = frame["W"], frame["H"]
    covered = frame["tri"] != sr.NO_TRIANGLE
    d = math.sqrt(12.0)
    radius = (H / 2) * (1.0 / math.sqrt(d * d - 1.0)) / math.tan(math.radians(22.5))
    area = math.pi * radius * radius
    assert abs(int(covered.sum()) - area) <= 4.0 * 2.0 * math.pi * radius * 0.5 + 0.002 * area   # within about two pixels of rim
    ys, xs = np.nonzero(covered)
    assert abs(xs.mean() - (W - 1) / 2) < 0.5 and abs(ys.mean() - (H - 1) / 2) < 0.5
    r = np.hypot(xs - (W - 1) / 2, ys - (H - 1) / 2)
    assert r.max() <= radius + 1.5
    yy, xx = np.mgrid[0:H, 0:W]
    inside = np.hypot(xx - (W - 1) / 2, yy - (H - 1) / 2) <= radius - 1.5
    assert covered[inside].all(), "holes inside the silhouette"
    depth = frame["depth"]
    assert (depth[~covered] == 1.0).all()
    dc = depth[covered]
    assert dc.min() > 0.0 and dc.max() < 1.0
    # depth of the nearest point of the sphere (frame centre): z_ndc of a point at view distance d-1, near .1, far 10
    zn, zf, dist = 0.1, 10.0, d - 1.0
    z_centre = zf * (dist - zn) / (dist * (zf - zn))
    assert abs(float(depth[H // 2, W // 2]) - z_centre) < 2e-4
    assert abs(float(dc.min()) - z_centre) < 2e-4


def test_second_draw_gives_the_same_frame(sr, device, frame):
    tgt = frame["tgt"]
    sr.draw(device, tgt, frame["mesh"], frame["ubo"], frame["tex"])
    assert np.array_equal(tgt.read_triangle_ids(0)[..., 0], frame["tri"])
    assert np.array_equal(tgt.read_depth(0)[..., 0].view(np.uint32), frame["depth"].view(np.uint32))
    assert np.array_equal(tgt.read_color(0), frame["bgra"])
    # (tile_refs may drop to 0: with the sparse-tail variant a repeated frame skips the tile lists)
    again = tgt.stats()
    assert {k: v for k, v in again.items() if k != "tile_refs"} == {k: v for k, v in frame["stats"].items() if k != "tile_refs"}
    assert again["tile_refs"] in (0, frame["stats"]["tile_refs"])


def test_screen_split_gives_the_same_frame(sr, device, frame):
    """Two bands with an odd split line and unequal heights, as a sort-first split over GPUs makes them."""
    W, H = frame["W"], frame["H"]
    cut = (H * 5) // 11 | 1
    for band in ((0, 0, W, cut), (0, cut, W, H - cut)):
        tgt = sr.RenderTarget(device, W, H, samples=1, n_views=1, rect=band)
        try:
            sr.draw(device, tgt, frame["mesh"], frame["ubo"], frame["tex"])
            rows = slice(band[1], band[1] + band[3])
            assert np.array_equal(tgt.read_triangle_ids(0)[..., 0], frame["tri"][rows])
            assert np.array_equal(tgt.read_depth(0)[..., 0].view(np.uint32), frame["depth"][rows].view(np.uint32))
            assert np.array_equal(tgt.read_color(0), frame["bgra"][rows])
        finally:
            tgt.destroy()


def test_triangle_order_does_not_matter(sr, device, frame):
    """Shuffle the triangle list: same depth buffer bit for bit, the same triangles win (their new ids
    map back to the old ones), the same colours."""
    T = frame["idx"].size // 3
    perm = np.random.default_rng(7).permutation(T)                 # new id k holds old triangle perm[k]
    idx2 = np.ascontiguousarray(frame["idx"].reshape(T, 3)[perm]).reshape(-1)
    mesh2 = sr.create_mesh_from_data(device, frame["verts"], idx2)
    tgt = sr.RenderTarget(device, frame["W"], frame["H"], samples=1, n_views=1)
    try:
        sr.draw(device, tgt, mesh2, frame["ubo"], frame["tex"])
        tri2, depth2, bgra2 = tgt.read_triangle_ids(0)[..., 0], tgt.read_depth(0)[..., 0], tgt.read_color(0)
    finally:
        tgt.destroy()
        mesh2.destroy()
    assert np.array_equal(depth2.view(np.uint32), frame["depth"].view(np.uint32))
    covered = frame["tri"] != sr.NO_TRIANGLE
    assert np.array_equal(tri2 != sr.NO_TRIANGLE, covered)
    # closed surface, back faces culled: no two front triangles overlap, so there are no depth ties to break by id
    # (a sample where snapping makes two of them overlap at exactly equal depth would go to the lower id: allow a handful)
    differ = perm[tri2[covered]] != frame["tri"][covered]
    assert int(differ.sum()) <= 16, f"{int(differ.sum())} samples went to another triangle"
    assert int((bgra2 != frame["bgra"]).any(axis=-1).sum()) <= 16


def _project(ubo, p):
    view = ubo[16:32].reshape(4, 4).T.astype(np.float64)
    proj = ubo[32:48].reshape(4, 4).T.astype(np.float64)
    c = proj @ view @ np.array([p[0], p[1], p[2], 1.0])
    return c[:2] / c[3]


def test_windows_of_the_frame_match_the_oracle(sr, oracle, frame):
    """Centre of the disc (smallest depth, magnified texture), the left rim (silhouette, grazing
    triangles), the north pole (degenerate and sliver triangles, minified texture) and the texture
    seam at u = 0/1: each window rasterised by the oracle from the same 10 M triangles."""
    W, H = frame["W"], frame["H"]
    ww, wh = (96, 64) if SMALL else (192, 128)
    d = math.sqrt(12.0)
    radius = (H / 2) * (1.0 / math.sqrt(d * d - 1.0)) / math.tan(math.radians(22.5))

    def window_at(cx, cy):
        x = int(min(max(cx - ww // 2, 0), W - ww)) | 1     # odd origin: 2x2 quads straddle the window edge
        y = int(min(max(cy - wh // 2, 0), H - wh)) | 1
        return (x, y, ww, wh)

    pole = _project(frame["ubo"], (0.0, 0.0, 1.0))
    seam = _project(frame["ubo"], (math.sqrt(0.5), 0.0, math.sqrt(0.5)))
    to_px = lambda ndc: ((ndc[0] * 0.5 + 0.5) * W, (-ndc[1] * 0.5 + 0.5) * H)
    windows = {"centre": window_at(W / 2, H / 2), "rim": window_at(W / 2 - radius, H / 2),
               "pole": window_at(*to_px(pole)), "seam": window_at(*to_px(seam))}
    chains = []
    for im in frame["images"]:
        if im is None:
            im = sr.scene.constant_rgba(4, 4, (128, 128, 255, 255))    # the normal map is bound but never sampled
        chains.append((oracle.build_mip_chain(im), im.shape[1], im.shape[0]))
    for name, (x, y, w, h) in windows.items():
        ref = oracle.draw(frame["ubo"], frame["verts"], frame["idx"], chains, W, H, rect=(x, y, w, h), threads=8)
        got_tri, got_depth = frame["tri"][y:y + h, x:x + w], frame["depth"][y:y + h, x:x + w]
        assert np.array_equal(got_tri, ref["tri"][..., 0]), f"{name}: triangle ids differ at {int((got_tri != ref['tri'][..., 0]).sum())} pixels"
        assert np.array_equal(got_depth.view(np.uint32), ref["depth"][..., 0].view(np.uint32)), f"{name}: depth bits differ"
        frac, worst = util.color_match_fraction(frame["bgra"][y:y + h, x:x + w], ref["bgra"], tol=1)
        # The colour bar (1 LSB on 99.9 % of the pixels) is a whole-frame figure.  Around the pole the
        # triangles are slivers a fraction of a pixel wide under a noise-like texture: the shading pass
        # recomputes their window-space vertices with the hardware reciprocal, a 1-ulp difference there
        # can move a vertex by 1/256 pixel, and on a sliver that shifts the level of detail enough to
        # change a pixel by several steps.  Measured on the CPU execution of the kernels with every
        # reciprocal moved by up to 1 / 2 / 4 ulp: 39 / 83 / 119 of the window's 24576 pixels.  So that
        # window alone is held to 99 %; it is 0.3 % of the frame.
        assert frac >= (0.99 if name == "pole" else 0.999), f"{name}: only {frac:.5f} of the pixels within 1 LSB (worst {worst})"
        assert (got_tri != sr.NO_TRIANGLE).any(), f"{name}: window misses the sphere"


def test_config2_64_views_in_one_draw(sr, oracle, device):
    """BASELINE config 2: the reference-sized mesh (88 x 88 sphere) at 1920x1080, 64 orbit cameras in one
    draw.  Eight of the views are checked against the oracle, all of them for a plausible silhouette."""
    W, H = (192, 108) if SMALL else (1920, 1080)
    verts, idx = sr.scene.uv_sphere(88, 88)
    images = sr.scene.material_textures(64 if SMALL else 1024)
    ubos = sr.scene.orbit_ubos(64, W, H)
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = [sr.create_texture(device, im) for im in images]
    tgt = sr.RenderTarget(device, W, H, samples=1, n_views=64)
    try:
        sr.draw(device, tgt, mesh, ubos, tex)
        chains = [(oracle.build_mip_chain(im), im.shape[1], im.shape[0]) for im in images]
        expected_area = math.pi * ((H / 2) * (1.0 / math.sqrt(11.0)) / math.tan(math.radians(22.5))) ** 2
        for v in range(64):
            tri = tgt.read_triangle_ids(v)[..., 0]
            covered = int((tri != sr.NO_TRIANGLE).sum())
            assert abs(covered - expected_area) < 0.03 * expected_area, (v, covered, expected_area)   # a faceted sphere, every camera at the same distance
            if v % 8 == 3:
                ref = oracle.draw(ubos[v], verts, idx, chains, W, H, threads=8)
                assert np.array_equal(tri, ref["tri"][..., 0]) and np.array_equal(tgt.read_depth(v).view(np.uint32), ref["depth"].view(np.uint32)), v
                frac, worst = util.color_match_fraction(tgt.read_color(v), ref["bgra"], tol=1)
                assert frac >= 0.999, (v, frac, worst)
        st = tgt.stats()
        assert st["triangles"] == 64 * (idx.size // 3) and st["pixels"] == 64 * W * H
    finally:
        tgt.destroy()
        mesh.destroy()
        for t in tex:
            t.destroy()


def test_config4_sort_first_bands_of_an_8k_frame(sr, device, frame):
    """BASELINE config 4: the 10 M-triangle scene at 7680x4320 cut into the eight horizontal bands a
    sort-first split over eight GPUs would use (softraster.distributed.split_bands), each band drawn by its
    own target as a rank would: together they are exactly the frame drawn in one piece."""
    from softraster import distributed as d
    W, H = (960, 540) if SMALL else (7680, 4320)
    ubo = sr.scene.default_camera_ubo(W, H)
    whole = sr.RenderTarget(device, W, H, samples=1, n_views=1)
    try:
        sr.draw(device, whole, frame["mesh"], ubo, frame["tex"])
        vis, bgra = whole.read_visibility(0), whole.read_color(0)
    finally:
        whole.destroy()
    assert (vis[..., 0] & 0xFFFFFFFF != sr.NO_TRIANGLE).any()
    bands = d.split_bands(W, H, 8)
    assert sum(b[3] for b in bands) == H
    for band in bands:
        tgt = sr.RenderTarget(device, W, H, samples=1, n_views=1, rect=band)
        try:
            sr.draw(device, tgt, frame["mesh"], ubo, frame["tex"])
            rows = slice(band[1], band[1] + band[3])
            assert np.array_equal(tgt.read_visibility(0), vis[rows]), band
            assert np.array_equal(tgt.read_color(0), bgra[rows]), band
        finally:
            tgt.destroy()


def test_largest_frame(sr, oracle, device):
    """The largest frame a target accepts, 16384 x 16384 (256 x 256 tiles, 14-bit pixel coordinates): the
    silhouette is the closed-form disc, and three windows -- frame centre, the last tile column at the
    disc's right rim, the far corner -- equal the oracle's."""
    N = 1024 if SMALL else 16384
    verts, idx = sr.scene.uv_sphere(88, 88)
    images = sr.scene.material_textures(64)
    # a wide lens (100 degrees) is not needed: move the camera close so that the sphere reaches the frame's edge columns
    eye = (1.45, 1.45, 1.45)
    ubo = sr.scene.default_camera_ubo(N, N, eye=eye)
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = [sr.create_texture(device, im) for im in images]
    tgt = sr.RenderTarget(device, N, N, samples=1, n_views=1)
    try:
        sr.draw(device, tgt, mesh, ubo, tex)
        tri = tgt.read_triangle_ids(0)[..., 0]
        bgra = tgt.read_color(0)
        st = tgt.stats()
        covered = tri != sr.NO_TRIANGLE
        assert st["pixels"] == N * N and st["shaded_pixels"] == int(covered.sum())
        d = math.sqrt(3 * 1.45 ** 2)
        radius = (N / 2) * (1.0 / math.sqrt(d * d - 1.0)) / math.tan(math.radians(22.5))
        assert radius > N / 2, "the test wants the sphere to run over the frame's edges"
        assert covered[N // 2, :].all() and covered[:, N // 2].all()        # the central row and column are inside the disc
        assert not covered[0, 0] and not covered[N - 1, N - 1]               # the corners are outside
        w = 64 if SMALL else 192
        chains = [(oracle.build_mip_chain(im), im.shape[1], im.shape[0]) for im in images]
        # where the silhouette crosses the last column: x = N-1, |y - N/2| = sqrt(radius^2 - (N/2)^2)
        y_cross = int(N / 2 - math.sqrt(radius * radius - (N / 2) ** 2))
        for name, (x, y) in {"centre": (N // 2 - w // 2, N // 2 - w // 2), "last column": (N - w, max(0, y_cross - w // 2)), "corner": (N - w, N - w)}.items():
            ref = oracle.draw(ubo, verts, idx, chains, N, N, rect=(x, y, w, w), threads=8)
            assert np.array_equal(tri[y:y + w, x:x + w], ref["tri"][..., 0]), name
            frac, worst = util.color_match_fraction(bgra[y:y + w, x:x + w], ref["bgra"], tol=1)
            assert frac >= 0.999, (name, frac, worst)
        assert (tri[y_cross - 2:y_cross + 2, N - 8:] != sr.NO_TRIANGLE).any() and (tri[max(0, y_cross - w // 2):y_cross + w // 2, N - w:] == sr.NO_TRIANGLE).any()
    finally:
        tgt.destroy()
        mesh.destroy()
        for t in tex:
            t.destroy()
