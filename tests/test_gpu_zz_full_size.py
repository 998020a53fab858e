"""Parity at the sizes the BASELINE configurations name: the 10 M-triangle sphere at 3840x2160 with a
4096^2 albedo map (the headline), 64 cameras at 1920x1080, the 7680x4320 frame in sort-first bands, a
batch of cameras on a 50 M-triangle mesh.  This file checks
  * whole frames against the oracle (a 3840x2160 frame takes the oracle a few tenths of a second on the
    GPU box's host cores): bit-exact visibility and <= 1 LSB colour on >= 99.9 % of the pixels,
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
    assert st["triangles_unprojectable"] == 0 and st["triangles_clipped"] == 0
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
    # (tile_refs may drop to 0: a repeated frame that left nothing pending is drawn lean, without the tile lists)
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


def _chains(sr, oracle, images):
    out = []
    for im in images:
        if im is None:
            im = sr.scene.constant_rgba(4, 4, (128, 128, 255, 255))    # the normal map is bound but never sampled
        out.append((oracle.build_mip_chain(im), im.shape[1], im.shape[0]))
    return out


def _assert_frame_equals_oracle(sr, got_tri, got_depth, got_bgra, ref, what, windows=None):
    """Visibility bit for bit over the whole frame, colour to the bar over the whole frame and, where given, inside
    each named window (x, y, w, h) as well."""
    rt, rd = ref["tri"][..., 0], ref["depth"][..., 0]
    assert np.array_equal(got_tri, rt), f"{what}: triangle ids differ at {int((got_tri != rt).sum())} pixels"
    assert np.array_equal(got_depth.view(np.uint32), rd.view(np.uint32)), f"{what}: depth bits differ"
    covered = got_tri != sr.NO_TRIANGLE
    d = np.abs(got_bgra.astype(np.int32) - ref["bgra"].astype(np.int32)).max(axis=-1)
    assert (d[~covered] == 0).all(), f"{what}: background differs"
    frac = float((d[covered] <= 1).mean()) if covered.any() else 1.0      # over the shaded pixels only: the background cannot dilute it
    assert frac >= 0.999, f"{what}: only {frac:.5f} of the shaded pixels within 1 LSB (worst {int(d.max())})"
    for name, (x, y, w, h) in (windows or {}).items():
        c = covered[y:y + h, x:x + w]
        assert c.any(), f"{what}, {name}: window misses the mesh"
        f = float((d[y:y + h, x:x + w][c] <= 1).mean())
        assert f >= 0.999, f"{what}, {name}: only {f:.5f} of the window's shaded pixels within 1 LSB"


def _windows(frame):
    """Centre of the disc (smallest depth, magnified texture), the left rim (silhouette, grazing triangles), the north
    pole (degenerate and sliver triangles a fraction of a pixel wide, minified texture) and the texture seam at
    u = 0/1: each must meet the colour bar on its own, not only the frame as a whole."""
    W, H = frame["W"], frame["H"]
    ww, wh = (96, 64) if SMALL else (192, 128)
    d = math.sqrt(12.0)
    radius = (H / 2) * (1.0 / math.sqrt(d * d - 1.0)) / math.tan(math.radians(22.5))

    def window_at(cx, cy):
        return (int(min(max(cx - ww // 2, 0), W - ww)), int(min(max(cy - wh // 2, 0), H - wh)), ww, wh)

    pole = _project(frame["ubo"], (0.0, 0.0, 1.0))
    seam = _project(frame["ubo"], (math.sqrt(0.5), 0.0, math.sqrt(0.5)))
    to_px = lambda ndc: ((ndc[0] * 0.5 + 0.5) * W, (-ndc[1] * 0.5 + 0.5) * H)
    return {"centre": window_at(W / 2, H / 2), "rim": window_at(W / 2 - radius, H / 2), "pole": window_at(*to_px(pole)), "seam": window_at(*to_px(seam))}


def test_whole_frame_matches_the_oracle(sr, oracle, frame):
    """BASELINE config 3: every pixel of the 3840x2160 frame of the 10 M-triangle mesh against the oracle's frame of the
    same inputs; the four windows named in _windows are also held to the colour bar one by one."""
    W, H = frame["W"], frame["H"]
    ref = oracle.draw(frame["ubo"], frame["verts"], frame["idx"], _chains(sr, oracle, frame["images"]), W, H, threads=16)
    _assert_frame_equals_oracle(sr, frame["tri"], frame["depth"], frame["bgra"], ref, "config 3", _windows(frame))
    assert frame["stats"]["shaded_pixels"] == ref["stats"]["shaded_px"]


def test_whole_frame_with_the_reference_sampler(sr, oracle, device, frame):
    """The same frame with the sampler the reference creates (maxAnisotropy = 16, src/texture.cpp:255-256) on both
    sides: same visibility, colour to the bar."""
    W, H = frame["W"], frame["H"]
    tex = [sr.create_texture(device, im, sampler=sr.default_sampler(16.0)) if im is not None else None for im in frame["images"]]
    tgt = sr.RenderTarget(device, W, H, samples=1, n_views=1)
    oracle.set_anisotropic(True)
    try:
        sr.draw(device, tgt, frame["mesh"], frame["ubo"], tex)
        bgra = tgt.read_color(0)
        assert np.array_equal(tgt.read_triangle_ids(0)[..., 0], frame["tri"])
        ref = oracle.draw(frame["ubo"], frame["verts"], frame["idx"], _chains(sr, oracle, frame["images"]), W, H, threads=16)
    finally:
        oracle.set_anisotropic(False)
        tgt.destroy()
        for t in tex:
            if t is not None:
                t.destroy()
    assert (bgra != frame["bgra"]).any(), "the anisotropic footprint changed nothing: the sampler was not applied"
    _assert_frame_equals_oracle(sr, frame["tri"], frame["depth"], bgra, ref, "config 3, maxAnisotropy 16", _windows(frame))


def test_config2_64_views_in_one_draw(sr, oracle, device):
    """BASELINE config 2: the reference-sized mesh (88 x 88 sphere) at 1920x1080, 64 orbit cameras in one
    draw.  Every view is checked against the oracle, and for a plausible silhouette."""
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
            ref = oracle.draw(ubos[v], verts, idx, chains, W, H, threads=16)
            _assert_frame_equals_oracle(sr, tri, tgt.read_depth(v)[..., 0], tgt.read_color(v), ref, f"config 2, view {v}")
        st = tgt.stats()
        assert st["triangles"] == 64 * (idx.size // 3) and st["pixels"] == 64 * W * H
    finally:
        tgt.destroy()
        mesh.destroy()
        for t in tex:
            t.destroy()


def test_config4_sort_first_bands_of_an_8k_frame(sr, oracle, device, frame):
    """BASELINE config 4: the 10 M-triangle scene at 7680x4320 cut into the eight horizontal bands a
    sort-first split over eight GPUs would use (softraster.distributed.split_bands), each band drawn by its
    own target as a rank would: together they are exactly the frame drawn in one piece, and that frame is the
    oracle's, pixel for pixel.  The same with the cost-balanced bands of softraster.distributed.balance_bands."""
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
    ref = oracle.draw(ubo, frame["verts"], frame["idx"], _chains(sr, oracle, frame["images"]), W, H, threads=16)
    _assert_frame_equals_oracle(sr, (vis[..., 0] & 0xFFFFFFFF).astype(np.uint32), (vis[..., 0] >> 32).astype(np.uint32).view(np.float32), bgra, ref, "config 4")
    bands = d.split_bands(W, H, 8)
    assert sum(b[3] for b in bands) == H
    covered_rows = (vis[..., 0] & 0xFFFFFFFF != sr.NO_TRIANGLE).sum(axis=1)
    balanced = d.balance_bands(W // 8 + covered_rows, W, 8)      # a constant per row (clear, colour write) plus the shaded pixels
    assert sum(b[3] for b in balanced) == H and balanced != bands
    for band in bands + balanced:
        tgt = sr.RenderTarget(device, W, H, samples=1, n_views=1, rect=band)
        try:
            sr.draw(device, tgt, frame["mesh"], ubo, frame["tex"])
            rows = slice(band[1], band[1] + band[3])
            assert np.array_equal(tgt.read_visibility(0), vis[rows]), band
            assert np.array_equal(tgt.read_color(0), bgra[rows]), band
        finally:
            tgt.destroy()


def test_config5_view_batch_of_a_50M_triangle_mesh(sr, oracle, device):
    """BASELINE config 5 on one GPU: a share of the camera batch (16 orbit cameras of a 1024-camera orbit, 1920x1080) of
    the 50 M-triangle mesh (UV sphere 5000 x 5000) in one draw -- which the scratch budget splits into several passes.
    Two of the views against the oracle pixel for pixel, all of them by the silhouette of the sphere and the counters."""
    SX, SY, W, H, n_views = (300, 300, 192, 108, 16) if SMALL else (5000, 5000, 1920, 1080, 16)
    verts, idx = sr.scene.uv_sphere(SX, SY)
    images = sr.scene.material_textures(64 if SMALL else 1024)
    ubos = sr.scene.orbit_ubos(n_views, W, H, first=37 * n_views, total=1024)     # rank 37's cameras of the 1024
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = [sr.create_texture(device, im) for im in images]
    tgt = sr.RenderTarget(device, W, H, samples=1, n_views=n_views)
    try:
        sr.draw(device, tgt, mesh, ubos, tex)
        st = tgt.stats()
        T = idx.size // 3
        assert T == 2 * SX * SY and st["triangles"] == n_views * T and st["vertices"] == n_views * verts.shape[0] and st["pixels"] == n_views * W * H
        assert st["triangles_unprojectable"] == 0 and st["triangles_clipped"] == 0
        expected_area = math.pi * ((H / 2) * (1.0 / math.sqrt(11.0)) / math.tan(math.radians(22.5))) ** 2
        shaded = 0
        chains = _chains(sr, oracle, images)
        for v in range(n_views):
            tri = tgt.read_triangle_ids(v)[..., 0]
            covered = tri != sr.NO_TRIANGLE
            shaded += int(covered.sum())
            assert abs(int(covered.sum()) - expected_area) < 0.01 * expected_area + 200, (v, int(covered.sum()), expected_area)
            ys, xs = np.nonzero(covered)
            assert abs(xs.mean() - (W - 1) / 2) < 1.0 and abs(ys.mean() - (H - 1) / 2) < 1.0, v
            assert int(tri[covered].max()) < T
            if v in (3, n_views - 1):     # one view of the first pass, one of the last
                ref = oracle.draw(ubos[v], verts, idx, chains, W, H, threads=16)
                _assert_frame_equals_oracle(sr, tri, tgt.read_depth(v)[..., 0], tgt.read_color(v), ref, f"config 5, view {v}")
        assert st["shaded_pixels"] == shaded
    finally:
        tgt.destroy()
        mesh.destroy()
        for t in tex:
            t.destroy()


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
