"""Parity of the HIP draw path (through the C ABI) against the CPU oracle on identical inputs.

Bar: clip coordinates, window-space vertex records, coverage, depth and triangle ids are bit-exact;
colour is within 1 LSB per 8-bit channel on >= 99.9 % of the pixels (the oracle uses libm powf/log2f,
the kernels use the GPU's exp2/log2 instructions)."""
import os
import sys

import numpy as np
import pytest

import util

pytestmark = pytest.mark.gpu

COLOR_TOL_LSB = 1
COLOR_MIN_FRACTION = 0.999


def _check_view(got, ref, check_color=True):
    util.assert_visibility_equal(got, ref)
    if check_color:
        frac, worst = util.color_match_fraction(got["bgra"], ref["bgra"], COLOR_TOL_LSB)
        assert frac >= COLOR_MIN_FRACTION, f"colour within {COLOR_TOL_LSB} LSB on {frac:.5f} of pixels (worst {worst})"


@pytest.mark.parametrize("segments", [32, 88])
def test_vertex_stage_bit_exact(sr, oracle, device, segments):
    verts, idx = sr.scene.uv_sphere(segments, segments)
    ubo = sr.scene.default_camera_ubo(800, 600)
    mesh = sr.create_mesh_from_data(device, verts, idx)
    clip, screen = mesh.vertex_stage(ubo, 800, 600)
    mesh.destroy()
    rclip, rscreen = oracle.vertex_stage(ubo, verts, 800, 600)
    assert np.array_equal(clip.view(np.uint32), rclip.view(np.uint32))
    assert np.array_equal(screen.view(np.uint8), rscreen.view(np.uint8))


def test_vertex_stage_general_matrices_and_unprojectable(sr, oracle, device):
    rng = np.random.default_rng(3)
    verts = np.zeros((5000, 8), np.float32)
    verts[:, :3] = rng.uniform(-6, 6, size=(5000, 3))
    model = np.eye(4, dtype=np.float32)
    model[:3, :3] = rng.normal(size=(3, 3)).astype(np.float32)
    model[:3, 3] = (0.3, -0.2, 0.1)
    ubo = sr.scene.default_camera_ubo(1920, 1080, model=model)
    mesh = sr.create_mesh_from_data(device, verts, np.zeros(0, np.uint32))
    clip, screen = mesh.vertex_stage(ubo, 1920, 1080)
    mesh.destroy()
    rclip, rscreen = oracle.vertex_stage(ubo, verts, 1920, 1080)
    assert (rscreen["rw"] == 0).sum() > 10, "the test must contain vertices behind the eye"
    assert np.array_equal(clip.view(np.uint32), rclip.view(np.uint32))
    assert np.array_equal(screen.view(np.uint8), rscreen.view(np.uint8))


@pytest.mark.parametrize("samples", [1, 4])
@pytest.mark.parametrize("segments", [32, 88])
def test_reference_frame_800x600(sr, oracle, device, segments, samples, anisotropic):
    """BASELINE config 1: one 800x600 frame of the bundled-model stand-in with the default camera."""
    verts, idx = sr.scene.uv_sphere(segments, segments)
    ubo = sr.scene.default_camera_ubo(800, 600)
    images = sr.scene.material_textures(256)
    got = util.gpu_render(sr, device, verts, idx, images, ubo, 800, 600, samples)[0]
    ref = oracle.draw(ubo, verts, idx, images, 800, 600, samples, threads=8)
    _check_view(got, ref)
    assert got["stats"]["shaded_pixels"] == ref["stats"]["shaded_px"]


@pytest.mark.parametrize("samples", [1, 4])
def test_triangle_soup_large_and_small_with_ties(sr, oracle, device, samples, anisotropic):
    verts, idx = util.random_soup(6000, seed=11, size=0.8)
    ubo = sr.scene.default_camera_ubo(640, 360)
    images = sr.scene.material_textures(64, seed=5)
    got = util.gpu_render(sr, device, verts, idx, images, ubo, 640, 360, samples)[0]
    ref = oracle.draw(ubo, verts, idx, images, 640, 360, samples, threads=8)
    _check_view(got, ref)


def test_huge_triangles_cover_many_tiles(sr, oracle, device, anisotropic):
    verts, idx = sr.scene.screen_quad(z=0.0, half=30.0, uv_scale=40.0)
    ubo = sr.scene.default_camera_ubo(1920, 1080)
    images = sr.scene.material_textures(128)
    got = util.gpu_render(sr, device, verts, idx, images, ubo, 1920, 1080, 1)[0]
    ref = oracle.draw(ubo, verts, idx, images, 1920, 1080, 1, threads=8)
    _check_view(got, ref)


def test_near_plane_crossing_is_clipped(sr, oracle, device, anisotropic):
    """depthClampEnable = VK_FALSE (src/main.cpp:586): a quad that reaches behind the eye is cut at the near plane and
    its visible part drawn, not dropped."""
    verts, idx = sr.scene.screen_quad(z=0.0, half=500.0)   # reaches behind the eye
    small, sidx = sr.scene.uv_sphere(16, 16)
    v = np.concatenate([verts, small]); i = np.concatenate([idx, sidx + 4]).astype(np.uint32)
    ubo = sr.scene.default_camera_ubo(320, 240)
    images = sr.scene.material_textures(32)
    got = util.gpu_render(sr, device, v, i, images, ubo, 320, 240, 1)[0]
    ref = oracle.draw(ubo, v, i, images, 320, 240, 1)
    _check_view(got, ref)
    assert ref["stats"]["tris_clip_culled"] == 0 and ref["stats"]["tris_clipped"] == 2
    assert got["stats"]["triangles_unprojectable"] == 0 and got["stats"]["triangles_clipped"] == 2
    assert (got["tri"] < 2).sum() > 10000    # the quad fills most of the frame


@pytest.mark.parametrize("samples", [1, 4])
def test_fragments_exactly_on_the_far_plane(sr, oracle, device, samples):
    """Depth test LESS against the cleared 1.0 (src/main.cpp:602-610, :921): a fragment whose depth comes out as exactly
    1.0 is not drawn.  Triangles lying in the far plane, large and small (tile lists and the direct path), with identity
    matrices so that every vertex has depth 1.0f exactly: the interpolated depth is 1.0 at some samples and one ulp
    below at others (the weights sum to one only to an ulp) -- which is which must agree with the oracle, sample
    for sample."""
    W, H = 96, 64
    eye = np.eye(4, dtype=np.float32)
    ubo = sr.scene.make_ubo(eye, eye, eye, (0.0, 0.0, 0.0))
    rng = np.random.default_rng(17)
    tris = [np.array([[-0.93, -0.81, 1.0], [0.87, -0.77, 1.0], [0.05, 0.9, 1.0]])]                       # most of the frame
    for _ in range(300):                                                                                # and a cloud of a-few-pixel triangles
        c = rng.uniform(-0.9, 0.9, size=2)
        d = rng.uniform(-0.06, 0.06, size=(3, 2))
        t = np.concatenate([c + d, np.ones((3, 1))], axis=1)
        if (t[1, 0] - t[0, 0]) * (t[2, 1] - t[0, 1]) - (t[2, 0] - t[0, 0]) * (t[1, 1] - t[0, 1]) < 0:   # counter-clockwise on the y-up image
            t = t[::-1]
        tris.append(t)
    verts = np.zeros((3 * len(tris), 8), np.float32)
    verts[:, :3] = np.concatenate(tris)
    verts[:, 5] = 1.0
    verts[:, 6:8] = verts[:, :2]
    idx = np.arange(3 * len(tris), dtype=np.uint32)
    images = sr.scene.material_textures(16)
    got = util.gpu_render(sr, device, verts, idx, images, ubo, W, H, samples)[0]
    ref = oracle.draw(ubo, verts, idx, images, W, H, samples)
    util.assert_visibility_equal(got, ref)
    drawn = ref["tri"] != sr.NO_TRIANGLE
    assert drawn.any() and not drawn.all() and (ref["depth"][drawn] < 1.0).all()
    # the case is really there: inside the big triangle some samples are drawn (one ulp below 1.0) and some are not (exactly 1.0)
    inside = ref["tri"][20:40, 40:56]
    assert (inside != sr.NO_TRIANGLE).any() and (inside == sr.NO_TRIANGLE).any()


def _window_triangles(sr, W, H, tris):
    """Triangles given in window coordinates (x, y in pixels, z in 0..1) under identity matrices; each is turned front-facing."""
    eye = np.eye(4, dtype=np.float32)
    ubo = sr.scene.make_ubo(eye, eye, eye, (0.0, 0.0, 0.0))
    out = []
    for t in tris:
        t = np.asarray(t, np.float64)
        if (t[1, 0] - t[0, 0]) * (t[2, 1] - t[0, 1]) - (t[2, 0] - t[0, 0]) * (t[1, 1] - t[0, 1]) >= 0:   # clockwise on the y-down window = front
            t = t[::-1]
        ndc = np.stack([t[:, 0] / (W / 2) - 1.0, 1.0 - t[:, 1] / (H / 2), t[:, 2]], axis=1)
        out.append(ndc)
    verts = np.zeros((3 * len(out), 8), np.float32)
    verts[:, :3] = np.concatenate(out)
    verts[:, 5] = 1.0
    verts[:, 6:8] = verts[:, :2]
    return ubo, verts, np.arange(3 * len(out), dtype=np.uint32)


@pytest.mark.parametrize("samples", [1, 4])
def test_tall_and_wide_slivers_then_another_scene_in_the_same_target(sr, oracle, device, samples, anisotropic):
    """Triangles one pixel wide and a few dozen rows tall (and the transpose) are resolved by the set-up kernel itself; their
    boxes reach over several of the 32 x 8-pixel blocks the shading pass and the next draw's clear go by.  Every block must be
    shaded in this draw and cleared by the next one: a second, unrelated scene (a large triangle BEHIND where the slivers were)
    is drawn into the same target and must show nothing of the first."""
    W, H = 160, 96
    slivers = [
        [[40.2, 5.0, 0.2], [41.4, 5.2, 0.2], [40.8, 29.0, 0.2]],       # 1 x 24 pixels: block rows 0..3
        [[90.3, 33.1, 0.3], [91.2, 33.0, 0.3], [90.7, 64.9, 0.3]],     # 1 x 31: block rows 4..8
        [[8.1, 70.3, 0.25], [39.9, 70.6, 0.25], [8.0, 71.4, 0.25]],    # 31 x 1: two block columns
        [[120.2, 10.0, 0.2], [122.3, 10.1, 0.2], [121.0, 25.9, 0.2]],  # 2 x 16
    ]
    if samples == 4:   # the direct path takes boxes of at most 8 pixels at four samples
        slivers = [[[40.1, 5.1, 0.2], [40.95, 5.2, 0.2], [40.6, 12.9, 0.2]], [[90.3, 30.1, 0.3], [90.9, 30.0, 0.3], [90.7, 37.9, 0.3]],
                   [[60.1, 70.3, 0.25], [67.9, 70.6, 0.25], [60.0, 70.9, 0.25]]]
    ubo, verts, idx = _window_triangles(sr, W, H, slivers)
    behind = [[[2.0, 2.0, 0.6], [158.0, 3.0, 0.6], [70.0, 94.0, 0.6]]]
    ubo2, verts2, idx2 = _window_triangles(sr, W, H, behind)
    images = sr.scene.material_textures(16)
    tex = [sr.create_texture(device, im) for im in images]
    m1, m2 = sr.create_mesh_from_data(device, verts, idx), sr.create_mesh_from_data(device, verts2, idx2)
    tgt = sr.RenderTarget(device, W, H, samples=samples)
    try:
        for mesh, (u, v, i) in ((m1, (ubo, verts, idx)), (m2, (ubo2, verts2, idx2)), (m1, (ubo, verts, idx)), (m1, (ubo, verts, idx))):
            sr.draw(device, tgt, mesh, u, tex)
            got = {"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}
            ref = oracle.draw(u, v, i, images, W, H, samples)
            util.assert_visibility_equal(got, ref)
            assert np.array_equal(got["bgra"][..., 3], ref["bgra"][..., 3])
            frac, worst = util.color_match_fraction(got["bgra"], ref["bgra"])
            assert frac >= 0.9995, (frac, worst)    # (a sliver is a few dozen pixels of the 15 360: an unshaded block must not pass)
            if mesh is m1:
                st = tgt.stats()
                assert st["triangles_direct"] == len(slivers) and st["triangles_binned"] == 0, st
                rows_covered = (ref["tri"].reshape(H, W, samples)[:, :, 0] != sr.NO_TRIANGLE).any(axis=1).sum()
                assert rows_covered >= (25 if samples == 1 else 6)    # (a sliver narrows towards its far end: not every row of its box holds a covered sample)
    finally:
        tgt.destroy(); m1.destroy(); m2.destroy()
        for t in tex:
            t.destroy()


def test_empty_mesh_and_offscreen_mesh(sr, oracle, device):
    images = sr.scene.material_textures(16)
    ubo = sr.scene.default_camera_ubo(128, 96)
    got = util.gpu_render(sr, device, np.zeros((0, 8), np.float32), np.zeros(0, np.uint32), images, ubo, 128, 96, 1)[0]
    assert (got["tri"] == sr.NO_TRIANGLE).all() and (got["depth"] == 1.0).all()
    assert (got["bgra"] == np.array([0, 0, 0, 255], np.uint8)).all()
    verts, idx = sr.scene.uv_sphere(8, 8)
    verts[:, 0] += 50.0
    got = util.gpu_render(sr, device, verts, idx, images, ubo, 128, 96, 1)[0]
    assert (got["tri"] == sr.NO_TRIANGLE).all()


@pytest.mark.parametrize("samples", [1, 4])
def test_target_is_reused_across_different_draws(sr, oracle, device, samples, anisotropic):
    """One target, a sequence of unrelated draws: nothing of an earlier frame may survive into a later
    one.  Sphere in two views -> nothing in view 0 only -> a small soup in both views -> the sphere again
    from another camera in view 0 only; every frame is checked against the oracle."""
    W, H = 200, 120
    sphere = sr.scene.uv_sphere(24, 16)
    soup = util.random_soup(40, 5, size=0.5)
    nothing = (np.zeros((0, 8), np.float32), np.zeros(0, np.uint32))
    images = sr.scene.material_textures(32)
    cams = sr.scene.orbit_ubos(3, W, H)
    tex = [sr.create_texture(device, im) for im in images]
    meshes = {k: sr.create_mesh_from_data(device, *v) for k, v in (("sphere", sphere), ("soup", soup), ("nothing", nothing))}
    data = {"sphere": sphere, "soup": soup, "nothing": nothing}
    tgt = sr.RenderTarget(device, W, H, samples=samples, n_views=2)
    try:
        expected = [None, None]   # what each view of the target must hold now
        for what, ubos in (("sphere", cams[:2]), ("nothing", cams[:1]), ("soup", cams[1:3]), ("sphere", cams[2:3])):
            sr.draw(device, tgt, meshes[what], ubos, tex)
            for v in range(len(ubos)):
                expected[v] = oracle.draw(ubos[v], data[what][0], data[what][1], images, W, H, samples)
            for v in range(2):
                got = {"tri": tgt.read_triangle_ids(v), "depth": tgt.read_depth(v), "bgra": tgt.read_color(v)}
                util.assert_visibility_equal(got, expected[v])
                frac, worst = util.color_match_fraction(got["bgra"], expected[v]["bgra"])
                assert frac >= 0.999, (what, v, frac, worst)
    finally:
        tgt.destroy()
        for m in meshes.values():
            m.destroy()
        for t in tex:
            t.destroy()


def _random_view(rng, W, H, sr):
    """Random model matrix (rotation, non-uniform scale, sometimes mirrored, small shift) and camera."""
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    a, b, c, d = q
    rot = np.array([[a*a+b*b-c*c-d*d, 2*(b*c-a*d), 2*(b*d+a*c)], [2*(b*c+a*d), a*a-b*b+c*c-d*d, 2*(c*d-a*b)], [2*(b*d-a*c), 2*(c*d+a*b), a*a-b*b-c*c+d*d]])
    scale = np.diag(rng.uniform(0.6, 1.6, size=3) * np.array([1.0, 1.0, rng.choice([1.0, 1.0, -1.0])]))
    model = np.eye(4, dtype=np.float32)
    model[:3, :3] = (rot @ scale).astype(np.float32)
    model[:3, 3] = rng.uniform(-0.3, 0.3, size=3)
    direction = rng.normal(size=3); direction /= np.linalg.norm(direction)
    eye = direction * rng.uniform(2.2, 7.0)
    up = (0.0, 0.0, 1.0) if abs(direction[2]) < 0.9 else (0.0, 1.0, 0.0)
    view = sr.scene.look_at(eye, rng.uniform(-0.4, 0.4, size=3), up)
    proj = sr.scene.perspective(np.radians(rng.uniform(20.0, 95.0)), W / H, 0.1, 20.0)
    return sr.scene.make_ubo(model, view, proj, eye)


def test_mesh_and_textures_destroyed_before_the_read_back(sr, oracle):
    """The reference may destroy a mesh and its textures as soon as the device is idle.  Here a draw whose tile
    reference list overflowed is re-issued when its target is next read -- which needs the mesh and the textures --
    so destroying them first must complete that draw: the read-back and the counters afterwards are the oracle's."""
    verts, idx = util.random_soup(300, 21, size=1.2)
    W, H = 320, 200
    ubo = sr.scene.default_camera_ubo(W, H)
    images = sr.scene.material_textures(32)
    dev = util.device_with_env(sr, SR_REF_CAPACITY="64")
    try:
        mesh = sr.create_mesh_from_data(dev, verts, idx)
        tex = [sr.create_texture(dev, im) for im in images]
        tgt = sr.RenderTarget(dev, W, H)
        sr.draw(dev, tgt, mesh, ubo, tex)
        mesh.destroy()
        for t in tex:
            t.destroy()
        junk = sr.create_mesh_from_data(dev, np.zeros((3 * 4096, 8), np.float32), np.arange(3 * 4096, dtype=np.uint32))   # likely to reuse the freed memory
        got = {"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}
        st = tgt.stats()
        junk.destroy()
        tgt.destroy()
        _check_view(got, oracle.draw(ubo, verts, idx, images, W, H))
        assert st["triangles"] == idx.size // 3 and st["vertices"] == verts.shape[0] and st["tile_refs"] > 64
    finally:
        dev.close()


FUZZ_SEEDS_THAT_NEED_EXACT_SHADING = (199, 427, 613, 1040, 1236, 1414)   # found by running the fuzzer on the former shading pass


@pytest.mark.parametrize("seed", FUZZ_SEEDS_THAT_NEED_EXACT_SHADING)
def test_vertex_near_the_eye_plane_and_far_off_frame(sr, oracle, device, seed):
    """Cases of the differential fuzzer (tests/emu/fuzz_parity.py) in which a visible triangle has a vertex far outside
    the frame and almost on the eye plane: its 1/w is huge, and a shading pass that takes one barycentric weight as one
    minus the other two, or recomputes the window-space vertices with an approximate reciprocal, colours such a triangle
    wrongly (up to 60 % of a frame's pixels off the bar).  Visibility, clip counts and colour must all meet the bar."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import fuzz_parity
    fuzz_parity.check_seed(seed, sr, oracle, device, util)


def test_a_block_of_fuzz_cases(sr, oracle, device):
    """160 cases of the differential fuzzer (random soups, dust, slivers and strips with duplicates and degenerate triangles,
    random cameras -- some inside the cloud --, frame sizes, sample counts, stored rectangles and view batches) through the C
    ABI on the GPU: visibility and clip counts equal the oracle's, colour within the bar on every well-covered frame.  The
    hardware's reciprocal, square root, exponential and logarithm instructions take part here; the CPU execution of the
    kernels can only approximate them."""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
    import fuzz_parity
    lowest = 1.0
    count = 12 if os.environ.get("SR_TEST_EMULATED") == "1" else 160     # (the CPU execution of the kernels has its own fuzz leg)
    for seed in range(7000, 7000 + count):
        lowest = min(lowest, fuzz_parity.check_seed(seed, sr, oracle, device, util, vertex_stage_too=seed % 4 == 0)[0])
    assert lowest >= 0.999


def test_lean_draws_and_their_fallback(sr, oracle, anisotropic):
    """A draw that follows one which left nothing for the tile lists and the clipper is issued without those kernels;
    if it turns out to need them after all, it is issued again in full before anything of it can be observed.  One
    target, one mesh, a camera that jumps: far (micro-triangles only) twice, close (triangles of a dozen pixels), far
    three times, then almost on the surface (vertices behind the eye: the clipper is needed).  Every frame and its
    counters equal the oracle's."""
    W, H = 192, 108
    verts, idx = sr.scene.uv_sphere(200, 150)
    images = sr.scene.material_textures(32)
    far, close, touching = (2.0, 2.0, 2.0), (0.65, 0.65, 0.63), (0.585, 0.58, 0.575)
    dev = sr.Device(0)
    try:
        mesh = sr.create_mesh_from_data(dev, verts, idx)
        tex = [sr.create_texture(dev, im) for im in images]
        tgt = sr.RenderTarget(dev, W, H)
        seen = []
        import math
        for eye in (far, far, close, far, far, far, touching, far):
            ubo = sr.scene.default_camera_ubo(W, H, eye=eye)
            if eye is touching:     # just above the surface, looking along it: part of the sphere lies behind the eye plane
                e = np.array([0.0, -1.02, 0.0])
                ubo = sr.scene.make_ubo(np.eye(4, dtype=np.float32), sr.scene.look_at(e, (0.9, -0.9, 0.1), (0.0, 0.0, 1.0)),
                                        sr.scene.perspective(math.radians(45.0), W / H, 0.002, 10.0), e)   # (a near plane closer than a triangle is wide)
            sr.draw(dev, tgt, mesh, ubo, tex)
            got = {"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}
            st = tgt.stats()
            ref = oracle.draw(ubo, verts, idx, images, W, H)
            _check_view(got, ref)
            assert st["triangles_clipped"] == ref["stats"]["tris_clipped"] and st["triangles_unprojectable"] == 0
            seen.append((st["triangles_binned"], st["triangles_clipped"]))
        assert seen[0] == seen[1] == (0, 0), seen          # the far camera leaves nothing pending: draw 2 was lean
        assert seen[2][0] > 0 and seen[6][1] > 0, seen     # draws 3 and 7 were issued lean and needed the full pipeline
        assert seen[5] == (0, 0) and seen[7][1] == 0, seen
        tgt.destroy()
        mesh.destroy()
        for t in tex:
            t.destroy()
    finally:
        dev.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_few_pending_triangles_take_the_direct_rasteriser(sr, oracle, samples, anisotropic):
    """When a target's previous draw left only a few thousand small triangles for the tile lists, the next draw of the same mesh
    resolves them with one kernel straight into the visibility buffer (no binning, scan, fill, tile rasteriser).  The reference's
    own frame drawn three times (tile path, then twice direct), two views at once, a stored rectangle, a camera close to the surface
    (vertices behind the eye: the clipper rides along), and a scene with one triangle far too large for that kernel (the draw is
    repeated through the tile path): every frame, and its counters, equal the oracle's."""
    import math
    W, H = 400, 300
    images = sr.scene.material_textures(64, seed=6)
    sphere = sr.scene.uv_sphere(88, 88)
    dev = sr.Device(0)
    try:
        tex = [sr.create_texture(dev, im) for im in images]
        mesh = sr.create_mesh_from_data(dev, *sphere)
        # one target, three draws; then a second target of two views and a third with a rectangle
        for n_views, rect in ((1, None), (2, None), (1, (37, 50, 301, 177))):
            tgt = sr.RenderTarget(dev, W, H, samples=samples, n_views=n_views, rect=rect)
            ubos = sr.scene.orbit_ubos(3, W, H)
            for k in range(3):
                before = dev.draw_counts()
                u = ubos[k:k + 1] if n_views == 1 else np.stack([ubos[k], ubos[(k + 1) % 3]])
                sr.draw(dev, tgt, mesh, u, tex)
                for v in range(n_views):
                    got = {"tri": tgt.read_triangle_ids(v), "depth": tgt.read_depth(v), "bgra": tgt.read_color(v)}
                    _check_view(got, oracle.draw(u[v], sphere[0], sphere[1], images, W, H, samples, rect=rect, threads=4))
                st, after = tgt.stats(), dev.draw_counts()
                assert st["triangles_binned"] > 100 * n_views
                assert after["direct_draws"] - before["direct_draws"] == (1 if k > 0 else 0), (n_views, rect, k)
                assert after["lean_redraws"] == before["lean_redraws"]
            tgt.destroy()
        # the eye just above the surface: part of the sphere is behind the eye plane, the clipper is needed; the second and third draw are direct
        e = np.array([0.0, -1.02, 0.0])
        close = sr.scene.make_ubo(np.eye(4, dtype=np.float32), sr.scene.look_at(e, (0.9, -0.9, 0.1), (0.0, 0.0, 1.0)),
                                  sr.scene.perspective(math.radians(45.0), W / H, 0.002, 10.0), e)
        tgt = sr.RenderTarget(dev, W, H, samples=samples)
        for k in range(3):
            sr.draw(dev, tgt, mesh, close, tex)
            got = {"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}
            ref = oracle.draw(close, sphere[0], sphere[1], images, W, H, samples, threads=4)
            _check_view(got, ref)
            st = tgt.stats()
            assert st["triangles_clipped"] == ref["stats"]["tris_clipped"] > 0 and st["triangles_unprojectable"] == 0
        tgt.destroy(); mesh.destroy()
        # small triangles and one that covers most of the frame: the direct draw is given up and repeated, and not tried again
        verts, idx = util.random_soup(900, 8, size=0.12)
        big = np.zeros((3, 8), np.float32)
        big[:, :3] = [[-2.5, 0.4, -1.4], [0.6, -2.4, -1.2], [0.3, 0.5, 1.9]]
        big[:, 5] = 1.0
        ubo = sr.scene.default_camera_ubo(W, H)
        if oracle.draw(ubo, big, np.arange(3, dtype=np.uint32), images, W, H, samples, shade=False)["stats"]["shaded_px"] == 0:
            big = big[::-1].copy()
        verts = np.concatenate([verts, big]); idx = np.concatenate([idx, np.arange(3, dtype=np.uint32) + 2700]).astype(np.uint32)
        mesh = sr.create_mesh_from_data(dev, verts, idx)
        tgt = sr.RenderTarget(dev, W, H, samples=samples)
        ref = oracle.draw(ubo, verts, idx, images, W, H, samples, threads=4)
        assert (ref["tri"] == 900).sum() > W * H * samples // 8      # the large triangle shows
        redraws = []
        for k in range(4):
            before = dev.draw_counts()
            sr.draw(dev, tgt, mesh, ubo, tex)
            _check_view({"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}, ref)
            after = dev.draw_counts()
            redraws.append((after["direct_draws"] - before["direct_draws"], after["lean_redraws"] - before["lean_redraws"]))
        assert redraws[0] == (0, 0) and redraws[2] == (0, 0) and redraws[3] == (0, 0), redraws     # (the second may or may not try, by the first's references per triangle)
        assert redraws[1] in ((0, 0), (1, 1)), redraws
        tgt.destroy(); mesh.destroy()
        for t in tex:
            t.destroy()
    finally:
        dev.close()


def test_model_and_textures_from_files(sr, oracle, device, tmp_path):
    """Model::load and create_texture_from_file: a glTF file with two meshes and PNG maps, written here,
    drawn through the file-taking entry points, equals the oracle's frame of the arrays they were made of."""
    import test_asset_files as files
    sphere = sr.scene.uv_sphere(20, 14)
    quad = sr.scene.screen_quad(z=-0.2, half=1.4, uv_scale=2.0)
    doc, blob = files.make_gltf([sphere, quad], interleave=True)
    gltf = files.write_gltf(tmp_path, "scene.glb", doc, blob, "glb")
    images = sr.scene.material_textures(32, seed=3)
    paths = []
    for k, im in enumerate(images):
        path = str(tmp_path / f"map{k}.png")
        if k >= 2:   # the reference's metallic / roughness maps are one-channel files
            files.write_png(path, files._pack_rows(im[..., 0].astype(np.int64), 8), 32, 32, 0, 8, filters=[4, 2, 1])
        else:
            files.write_png(path, files._pack_rows(im.reshape(32, 32 * 4).astype(np.int64), 8), 32, 32, 6, 8, filters=[3, 4])
        paths.append(path)
    model = sr.Model.load(device, gltf)
    tex = [sr.create_texture_from_file(device, p) for p in paths]
    W, H = 240, 160
    ubo = sr.scene.default_camera_ubo(W, H)
    try:
        assert [t.width for t in tex] == [32] * 4 and tex[0].mip_levels == 6
        meshes = model.meshes()
        assert len(meshes) == 2 and meshes[0].indices_size == quad[1].size and meshes[1].indices_size == sphere[1].size   # root node first
        for mesh, (verts, idx) in zip(meshes, (quad, sphere)):
            tgt = sr.RenderTarget(device, W, H)
            try:
                sr.draw(device, tgt, mesh, ubo, tex)
                got = {"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}
            finally:
                tgt.destroy()
            _check_view(got, oracle.draw(ubo, verts, idx, images, W, H))
    finally:
        model.destroy()
        for t in tex:
            t.destroy()


def test_multi_view_batch_matches_single_views(sr, oracle, device, anisotropic):
    """BASELINE config 2 in small: a batch of orbit cameras drawn by one call."""
    verts, idx = sr.scene.uv_sphere(32, 32)
    ubos = sr.scene.orbit_ubos(6, 480, 270)
    images = sr.scene.material_textures(128)
    got = util.gpu_render(sr, device, verts, idx, images, ubos, 480, 270, 1)
    for k in range(6):
        ref = oracle.draw(ubos[k], verts, idx, images, 480, 270, 1, threads=4)
        _check_view(got[k], ref)


@pytest.mark.parametrize("samples", [1, 4])
def test_tile_reference_list_grows_and_the_draw_is_repeated(sr, oracle, samples):
    """The per-tile reference list is sized from earlier frames; a frame that needs more overflows it (the
    overflowing references are dropped and counted, never written) and is drawn again with a larger list
    when its target is next read.  Forced here with a list of 64 entries against thousands needed."""
    import os
    verts, idx = util.random_soup(300, 21, size=1.2)
    W, H = 320, 200
    ubos = sr.scene.orbit_ubos(2, W, H)
    images = sr.scene.material_textures(32)
    dev = util.device_with_env(sr, SR_REF_CAPACITY="64")
    try:
        for attempt in range(2):   # the second frame finds the list already large enough
            got = util.gpu_render(sr, dev, verts, idx, images, ubos, W, H, samples)
            assert got[0]["stats"]["tile_refs"] > 64 or got[0]["stats"]["tile_refs"] == 0
            for k in range(2):
                _check_view(got[k], oracle.draw(ubos[k], verts, idx, images, W, H, samples))
    finally:
        dev.close()


def _inward_sphere(sr, x_segments, y_segments, radius):
    """The UV sphere seen from inside: scaled, triangles turned round."""
    verts, idx = sr.scene.uv_sphere(x_segments, y_segments)
    verts = verts.copy()
    verts[:, :3] *= radius
    verts[:, 3:6] *= -1.0
    return verts, np.ascontiguousarray(idx.reshape(-1, 3)[:, ::-1]).reshape(-1)


@pytest.mark.parametrize("samples", [1, 4])
def test_near_plane_clipper(sr, oracle, samples, anisotropic):
    """Triangles with a vertex behind the eye are cut at the near plane and drawn, as the fixed-function clipper of
    the reference's pipeline does (depthClampEnable = VK_FALSE, src/main.cpp:586).  A ground
    quad that passes under and behind the camera, and a sphere seen from inside with the eye close to its surface
    (dozens of triangles reach from behind the eye to beyond the near plane, sharing their cut edges): visibility equals the oracle's
    clipper bit for bit, colour to the usual bar, in a three-view draw; then a clip list of 8 entries, where the
    triangles that find no room are dropped and counted."""
    import math
    W, H = 256, 160
    images = sr.scene.material_textures(64, seed=4)
    ground = np.zeros((4, 8), np.float32)
    ground[:, :3] = [[-2000, -2000, 0], [2000, -2000, 0], [2000, 2000, 0], [-2000, 2000, 0]]   # its cuts at the near plane leave the guard band
    ground[:, 5] = 1.0
    ground[:, 6:8] = ground[:, :2] * 0.1
    ball, ball_idx = _inward_sphere(sr, 32, 20, 1.0)     # triangles larger than the near distance: some span it
    ball[:, 2] += 1.0                                   # resting on the ground, the eye inside it
    verts = np.concatenate([ground, ball])
    idx = np.concatenate([np.array([0, 1, 2, 0, 2, 3], np.uint32), ball_idx + 4]).astype(np.uint32)

    def camera(eye, target):
        return sr.scene.make_ubo(np.eye(4, dtype=np.float32), sr.scene.look_at(eye, target, (0.0, 0.0, 1.0)),
                                 sr.scene.perspective(math.radians(45.0), W / H, 0.1, 10.0), eye)
    ubos = np.stack([camera(np.array([0.0, -0.97, 1.0]), (0.9, -0.5, 1.0)),     # 0.03 from the shell, looking along it
                     camera(np.array([0.2, 0.1, 1.2]), (0.0, 1.0, 0.3)),        # well inside: nothing of the ball is clipped
                     camera(np.array([0.3, -2.0, 1.0]), (0.0, 3.0, 0.0))])      # outside: the ground passes under the camera
    dev = sr.Device(0)
    try:
        got = util.gpu_render(sr, dev, verts, idx, images, ubos, W, H, samples)
        clipped = 0
        for k in range(3):
            ref = oracle.draw(ubos[k], verts, idx, images, W, H, samples, threads=4)
            _check_view(got[k], ref)
            assert ref["stats"]["tris_clip_culled"] == 0
            clipped += ref["stats"]["tris_clipped"]
        assert clipped > 40 and got[0]["stats"]["triangles_clipped"] == clipped and got[0]["stats"]["triangles_unprojectable"] == 0
        assert (got[0]["tri"] != sr.NO_TRIANGLE).mean() > 0.9 and (got[2]["tri"] < 2).sum() > W * H // 4
    finally:
        dev.close()
    dev = util.device_with_env(sr, SR_CLIP_CAPACITY="8")
    try:
        st = util.gpu_render(sr, dev, verts, idx, images, ubos[:1], W, H, samples)[0]["stats"]
        ref = oracle.draw(ubos[0], verts, idx, images, W, H, samples, shade=False)["stats"]
        assert st["triangles_clipped"] <= 8 and st["triangles_clipped"] + st["triangles_unprojectable"] >= ref["tris_clipped"] > 8
    finally:
        dev.close()


def test_more_distinct_tiles_than_a_binning_block_can_hash(sr, oracle):
    """One binning workgroup (1024 consecutive triangles) keeps its (triangle, tile) pairs in a 2048-slot hash
    table.  1024 small triangles, each sitting on a different tile corner (so 4096 distinct pairs over
    4096 distinct tiles), overflow it; the pairs that find no slot take the per-tile slow path.  16-pixel
    tiles (SR_TILE_LOG2=4, 4 samples) keep the frame small."""
    import os
    W = H = 2048
    T = 16
    per_row = (W // T - 2) // 2
    n = 1024
    assert (n // per_row + 1) * 2 * T + 2 * T <= H
    verts = np.zeros((n * 3, 8), np.float32)
    for k in range(n):
        cx, cy = T * (2 + 2 * (k % per_row)), T * (2 + 2 * (k // per_row))          # a tile corner, in pixels
        pts = [(cx - 3.3, cy - 2.6), (cx - 2.1, cy + 3.7), (cx + 3.4, cy + 0.4)]     # counter-clockwise as displayed
        for j, (x, y) in enumerate(pts):
            verts[3 * k + j, 0:3] = (2.0 * x / W - 1.0, 1.0 - 2.0 * y / H, 0.3 + 0.0001 * k)
            verts[3 * k + j, 5] = 1.0
            verts[3 * k + j, 6:8] = (x / 7.0, y / 5.0)
    idx = np.arange(n * 3, dtype=np.uint32)
    eye = np.eye(4, dtype=np.float32)
    ubo = sr.scene.make_ubo(eye, eye, eye, (0.0, 0.0, 5.0))
    images = sr.scene.material_textures(16)
    dev = util.device_with_env(sr, SR_TILE_LOG2="4")
    try:
        got = util.gpu_render(sr, dev, verts, idx, images, ubo, W, H, 4)[0]
        st = got["stats"]
        assert st["triangles_binned"] == n and (st["tile_refs"] == 4 * n or st["tile_refs"] == 0), st   # every triangle straddles 2 x 2 tiles
        _check_view(got, oracle.draw(ubo, verts, idx, images, W, H, 4, threads=4))
    finally:
        dev.close()


def test_both_tile_sizes_at_one_sample_give_the_same_frame(sr, oracle, device):
    """At one sample the tile rasteriser works on 32 x 32 tiles, and on 64 x 64 ones (SR_TILE_LOG2=6) for frames of more than 8192
    pixels along an axis: both must give the oracle's frame -- here a few views of large triangles, which all take the tile path."""
    verts, idx = sr.scene.uv_sphere(24, 16)
    n_views, W, H = 3, 400, 300
    ubos = sr.scene.orbit_ubos(n_views, W, H)
    images = sr.scene.material_textures(32)
    dev = util.device_with_env(sr, SR_TILE_LOG2="6")
    try:
        large = util.gpu_render(sr, dev, verts, idx, images, ubos, W, H, 1, draws=1)      # (a repeat of the draw could leave the tile path)
        small = util.gpu_render(sr, device, verts, idx, images, ubos, W, H, 1, draws=1)
        assert large[0]["stats"]["tile_refs"] > 0 and small[0]["stats"]["tile_refs"] >= large[0]["stats"]["tile_refs"]   # (equal when the whole run is made with the larger tiles)
        for k in range(n_views):
            ref = oracle.draw(ubos[k], verts, idx, images, W, H, 1)
            _check_view(large[k], ref)
            _check_view(small[k], ref)
            assert np.array_equal(large[k]["vis"], small[k]["vis"]) and np.array_equal(large[k]["bgra"], small[k]["bgra"])
    finally:
        dev.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_view_batch_split_into_passes(sr, oracle, device, samples):
    """BASELINE config 5 in small: more views than the per-view scratch budget holds at once, so one draw
    runs as several passes (here 1 MiB: a handful of views per pass).  Every view equals the oracle's, and
    the counters add up over the passes."""
    import os
    verts, idx = sr.scene.uv_sphere(40, 30)
    n_views, W, H = 13, 200, 120
    ubos = sr.scene.orbit_ubos(n_views, W, H)
    images = sr.scene.material_textures(32)
    dev = util.device_with_env(sr, SR_SCRATCH_MIB="1")
    try:
        got = util.gpu_render(sr, dev, verts, idx, images, ubos, W, H, samples, draws=1)   # (a repeat without tile lists needs less scratch)
        for k in range(n_views):
            ref = oracle.draw(ubos[k], verts, idx, images, W, H, samples)
            _check_view(got[k], ref)
        st = got[0]["stats"]
        assert st["triangles"] == n_views * (idx.size // 3) and st["pixels"] == n_views * W * H
        assert st["shaded_pixels"] == sum(int((g["tri"] != sr.NO_TRIANGLE).any(axis=-1).sum()) for g in got)
        # the same batch in one pass (default budget) reports the same work
        one = util.gpu_render(sr, device, verts, idx, images, ubos, W, H, samples, draws=1)
        assert st["passes"] > 1
        assert {k: v for k, v in one[0]["stats"].items() if k != "passes"} == {k: v for k, v in st.items() if k != "passes"}
        for a, b in zip(got, one):
            assert np.array_equal(a["vis"], b["vis"]) and np.array_equal(a["bgra"], b["bgra"])
    finally:
        dev.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_normal_mapping(sr, oracle, device, samples, anisotropic):
    """The normal map read (shaders/shader.frag:35-53 with the body the reference has commented out restored), an option that
    is off by default: same visibility, colour to the bar against the oracle with the same switch; switched off again the frame
    is the default one.  A second view with a general model matrix and a lean-eligible repeat of the draw are in the batch."""
    verts, idx = sr.scene.uv_sphere(64, 48)
    W, H = 400, 300
    rng = np.random.default_rng(5)
    ubos = np.stack([sr.scene.default_camera_ubo(W, H), _random_view(rng, W, H, sr)])
    images = sr.scene.material_textures(64, seed=3)
    images[1] = sr.scene.noise_rgba(64, 64, 9)
    plain = util.gpu_render(sr, device, verts, idx, images, ubos, W, H, samples)
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = [sr.create_texture(device, im) for im in images]
    tgt = sr.RenderTarget(device, W, H, samples=samples, n_views=2)
    device.set_normal_mapping(True)
    oracle.set_normal_mapping(True)
    try:
        for _ in range(2):   # (the second draw of the target would be a lean one if normal mapping allowed it)
            sr.draw(device, tgt, mesh, ubos, tex)
            for k in range(2):
                got = {"tri": tgt.read_triangle_ids(k), "depth": tgt.read_depth(k), "bgra": tgt.read_color(k)}
                _check_view(got, oracle.draw(ubos[k], verts, idx, images, W, H, samples))
                covered = (got["tri"] != sr.NO_TRIANGLE).any(axis=-1)
                changed = np.abs(got["bgra"].astype(int) - plain[k]["bgra"].astype(int)).max(axis=-1) > 4
                assert changed[covered].mean() > 0.25, f"the normal map changes {changed[covered].mean():.3f} of the covered pixels"
        with pytest.raises(sr.SoftRasterError):
            sr.draw(device, tgt, mesh, ubos, [tex[0], None, tex[2], tex[3]])
        device.set_normal_mapping(False)
        sr.draw(device, tgt, mesh, ubos, tex)   # switched off again: the default frame
        for k in range(2):
            assert np.array_equal(tgt.read_visibility(k), plain[k]["vis"]) and np.array_equal(tgt.read_color(k), plain[k]["bgra"])
    finally:
        device.set_normal_mapping(False)
        oracle.set_normal_mapping(False)
        tgt.destroy(); mesh.destroy()
        for t in tex:
            t.destroy()


@pytest.mark.parametrize("samples", [1, 4])
def test_workgroups_that_take_every_view_of_a_pass(sr, oracle, device, samples):
    """A large mesh drawn for a batch of views: the vertex and set-up kernels load positions and indices once and go
    through the views of the pass one after the other.  SR_FILL_BLOCKS=1 makes them do so for a small mesh too (by
    default its views would be spread over more workgroups).  11 views (more than travel as kernel arguments, so the
    batch form of the vertex kernel runs) of a soup with micro-triangles, large triangles and near-plane crossings."""
    verts, idx = util.random_soup(4000, seed=23, size=0.3)
    n_views, W, H = 11, 200, 120
    ubos = sr.scene.orbit_ubos(n_views, W, H)
    ubos[3] = sr.scene.default_camera_ubo(W, H, eye=(0.1, 0.05, 0.02))   # inside the cloud: triangles through the near plane
    images = sr.scene.material_textures(16)
    dev = util.device_with_env(sr, SR_FILL_BLOCKS="1")
    try:
        got = util.gpu_render(sr, dev, verts, idx, images, ubos, W, H, samples)
        for k in range(n_views):
            ref = oracle.draw(ubos[k], verts, idx, images, W, H, samples)
            _check_view(got[k], ref, check_color=int((ref["tri"] != sr.NO_TRIANGLE).any(axis=-1).sum()) >= 2000)
        spread = util.gpu_render(sr, device, verts, idx, images, ubos, W, H, samples)
        assert spread[0]["stats"] == got[0]["stats"]
        for a, b in zip(got, spread):
            assert np.array_equal(a["vis"], b["vis"]) and np.array_equal(a["bgra"], b["bgra"])
    finally:
        dev.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_screen_split_is_partition_independent(sr, oracle, device, samples):
    """Sort-first building block: a target restricted to a rectangle stores exactly the crop of the
    full frame, for rectangles that cut through tiles and pixels quads."""
    verts, idx = sr.scene.uv_sphere(64, 48)
    ubo = sr.scene.default_camera_ubo(640, 480)
    images = sr.scene.material_textures(128)
    full = util.gpu_render(sr, device, verts, idx, images, ubo, 640, 480, samples)[0]
    for rect in [(0, 0, 640, 240), (0, 240, 640, 240), (100, 51, 333, 207), (320, 0, 320, 480)]:
        x, y, w, h = rect
        part = util.gpu_render(sr, device, verts, idx, images, ubo, 640, 480, samples, rect=rect)[0]
        for key in ("tri", "depth", "bgra"):
            assert np.array_equal(part[key], full[key][y:y + h, x:x + w]), f"{key} differs in rect {rect}"
        ref = oracle.draw(ubo, verts, idx, images, 640, 480, samples, rect=rect, threads=4)
        _check_view(part, ref)


def test_chunks_outside_a_band_are_skipped_and_the_band_stays_the_same(sr, oracle):
    """A one-view draw into a target restricted to a rectangle (a band of a frame split over several GPUs) tests every chunk of 768
    consecutive triangles against the rectangle and skips the chunks -- and the vertices only they use -- that cannot touch it.
    Bands and odd rectangles of four scenes (a sphere under the default camera; the same under a rotating, scaling, mirroring model
    matrix; the eye inside a large sphere, where chunks reach behind the eye and must be kept; a soup whose shuffled triangle order
    makes every chunk refer to vertices all over the mesh) are bit-identical to the same draw without culling and to the oracle's
    rectangle; chunks are in fact culled; a whole frame drawn through the same device afterwards is untouched by it."""
    W, H = 480, 360
    rng = np.random.default_rng(9)
    sphere = sr.scene.uv_sphere(240, 160)                       # 76 800 triangles: 100 chunks
    big = _inward_sphere(sr, 200, 150, 3.0)
    soup_v, soup_i = util.random_soup(9000, 4, size=0.3)
    soup_i = soup_i.reshape(-1, 3)[rng.permutation(9000)].reshape(-1).astype(np.uint32)
    images = sr.scene.material_textures(64, seed=1)
    scenes = [(sphere, sr.scene.default_camera_ubo(W, H)), (sphere, _random_view(np.random.default_rng(5), W, H, sr)),
              (big, sr.scene.default_camera_ubo(W, H, eye=(0.4, 0.3, 0.2))), ((soup_v, soup_i), sr.scene.default_camera_ubo(W, H))]
    rects = [(0, 0, W, 90), (0, 90, W, 90), (0, 180, W, 64), (0, 244, W, 116), (130, 77, 201, 55), (0, 0, 31, H), (400, 300, 80, 60)]
    culling, plain = util.device_with_env(sr, SR_CULL_MIN_CHUNKS="1"), util.device_with_env(sr, SR_NO_CULL="1")
    try:
        for k, ((verts, idx), ubo) in enumerate(scenes):
            before = culling.draw_counts()
            for rect in rects:
                a = util.gpu_render(sr, culling, verts, idx, images, ubo, W, H, 1, rect=rect)[0]
                b = util.gpu_render(sr, plain, verts, idx, images, ubo, W, H, 1, rect=rect)[0]
                assert np.array_equal(a["vis"], b["vis"]) and np.array_equal(a["bgra"], b["bgra"]), (k, rect)
                path_dependent = ("passes", "tile_refs")      # (a repeat of a draw takes the tile path or not by counters that may not have arrived yet)
                assert {x: y for x, y in a["stats"].items() if x not in path_dependent} == {x: y for x, y in b["stats"].items() if x not in path_dependent}, (k, rect)
                _check_view(a, oracle.draw(ubo, verts, idx, images, W, H, 1, rect=rect, threads=4))
            after = culling.draw_counts()
            tested, culled = after["chunks_tested"] - before["chunks_tested"], after["chunks_culled"] - before["chunks_culled"]
            chunks = -(-(idx.size // 3) // 768)
            assert tested >= len(rects) * chunks, (k, tested)       # (every draw of a rectangle was tested; a redrawn one twice)
            if k in (0, 1):
                assert culled > tested // 3, (k, tested, culled)    # a band holds a fraction of the sphere
            whole = util.gpu_render(sr, culling, verts, idx, images, ubo, W, H, 1)[0]
            assert culling.draw_counts()["chunks_tested"] == after["chunks_tested"]     # the whole frame is not culled
            _check_view(whole, oracle.draw(ubo, verts, idx, images, W, H, 1, threads=4))
        assert plain.draw_counts()["chunks_tested"] == 0
    finally:
        culling.close(); plain.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_lean_view_batches_store_no_depth_plane(sr, oracle, samples):
    """A lean draw of more views than travel as kernel arguments stores no depth plane in its vertex stage; set-up works out the depths
    of the triangles that cover a sample again, with the vertex stage's own operations.  Eleven views of a sphere of sub-pixel
    triangles, drawn three times from moving cameras (the second and third draw are lean): workgroups that take one view each, workgroups
    that loop over the views, a scratch budget that splits the batch into passes -- every view equals the oracle's and the same draw by
    a device that keeps the plane, bit for bit."""
    W, H = (192, 108) if samples == 1 else (96, 54)     # (at four samples a micro-triangle's box holds 8 pixels at most)
    n_views = 11
    verts, idx = sr.scene.uv_sphere(200, 150)
    half = verts[::2, :3]
    half[half == 0.0] = -0.0          # (zeros of both signs among the coordinates)
    verts[::2, :3] = half
    images = sr.scene.material_textures(32, seed=2)
    devices = {"default": util.device_with_env(sr), "looping": util.device_with_env(sr, SR_FILL_BLOCKS="1"),
               "passes": util.device_with_env(sr, SR_SCRATCH_MIB="2"), "stored": util.device_with_env(sr, SR_STORE_DEPTH="1"),
               "looping without lists": util.device_with_env(sr, SR_FILL_BLOCKS="1", SR_NO_CHUNK_VERTS="1")}
    try:
        tex = {k: [sr.create_texture(d, im) for im in images] for k, d in devices.items()}
        mesh = {k: sr.create_mesh_from_data(d, verts, idx) for k, d in devices.items()}
        tgt = {k: sr.RenderTarget(d, W, H, samples=samples, n_views=n_views) for k, d in devices.items()}
        cams = sr.scene.orbit_ubos(n_views + 2, W, H)
        c, s_ = np.cos(0.7), np.sin(0.7)
        turned = np.array([[0.9 * c, -0.9 * s_, 0, 0.05], [0.9 * s_, 0.9 * c, 0, -0.02], [0, 0, 0.9, 0.03], [0, 0, 0, 1]], np.float32)   # a model matrix that is not the identity
        for k in range(3):
            u = cams[k:k + n_views].copy()
            for v in (3, 7):
                u[v] = sr.scene.make_ubo(turned, sr.scene.look_at((2.0, 2.0, 2.0), (0.0, 0.0, 0.0), (0.0, 0.0, 1.0)), sr.scene.perspective(np.radians(45.0), W / H, 0.1, 10.0), (2.0, 2.0, 2.0))
            for name, d in devices.items():
                sr.draw(d, tgt[name], mesh[name], u, tex[name])
            got = {name: [{"tri": tgt[name].read_triangle_ids(v), "depth": tgt[name].read_depth(v), "bgra": tgt[name].read_color(v)} for v in range(n_views)]
                   for name in devices}
            for v in range(n_views):
                _check_view(got["default"][v], oracle.draw(u[v], verts, idx, images, W, H, samples, threads=4))
                for name in ("looping", "passes", "stored", "looping without lists"):
                    for key in ("tri", "depth", "bgra"):
                        assert np.array_equal(got[name][v][key], got["default"][v][key]), (k, v, name, key)
            stats = {name: tgt[name].stats() for name in devices}
            assert stats["default"]["triangles_binned"] == 0 and stats["default"]["triangles_clipped"] == 0      # nothing pending: the repeats are lean
            assert stats["passes"]["passes"] > 1
            for name in ("looping", "stored", "looping without lists"):       # (passes: a pass without the depth plane holds more views)
                assert {x: y for x, y in stats[name].items() if x != "passes"} == {x: y for x, y in stats["default"].items() if x != "passes"}, (k, name)
        for name, d in devices.items():
            c = d.draw_counts()
            assert c["lean_draws"] == 2 and c["lean_redraws"] == 0, (name, c)
            assert c["chunk_list_draws"] == (0 if name == "looping without lists" else 3), (name, c)
    finally:
        for d in devices.values():
            d.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_chunk_vertex_lists(sr, oracle, samples):
    """Set-up workgroups that take several views of a batch load the distinct vertices of their 768 triangles once per view, through
    lists made when the mesh is created.  Meshes at the limits of the lists: an indexed sphere whose last chunk is a few triangles
    long; the same triangles in shuffled order with shuffled vertex numbers (chunks of up to 1024 distinct vertices: lists exist only
    if every chunk fits); vertex numbers spread over more than 65536 (no lists); a triangle soup (2304 distinct vertices per chunk: no
    lists); chunks whose vertices straddle the span limit by one.  Every view equals the oracle's and the same draw without the lists,
    bit for bit, on the first (full) and on the repeated (lean) draws, with and without the stored depth plane."""
    W, H = (160, 90) if samples == 1 else (96, 54)
    n_views = 9
    rng = np.random.default_rng(5)
    verts, idx = sr.scene.uv_sphere(90, 70)
    tris = idx.reshape(-1, 3)
    meshes = {"indexed": (verts, idx, True)}
    renumber = rng.permutation(len(verts)).astype(np.uint32)                 # vertex numbers all over a range of < 65536
    shuffled_verts = np.empty_like(verts); shuffled_verts[renumber] = verts
    meshes["renumbered"] = (shuffled_verts, renumber[tris[rng.permutation(len(tris))]].reshape(-1), False)
    soup_verts = verts[tris.reshape(-1)].copy()
    meshes["soup"] = (soup_verts, np.arange(len(soup_verts), dtype=np.uint32), False)
    for gap, fits in ((65535 - int(tris[:768].max()), True), (65536 - int(tris[:768].max()), False)):
        # the first chunk also uses one vertex `gap` numbers above its highest: a span of exactly 65536 vertex numbers fits, one more does not
        far = int(tris[:768].max()) + gap
        wide_verts = np.zeros((max(far + 1, len(verts)), verts.shape[1]), verts.dtype)
        wide_verts[:len(verts)] = verts
        wide_verts[far] = verts[int(tris[0, 0])]
        wide_tris = tris.copy(); wide_tris[0, 0] = far
        meshes["span %d" % (far - int(wide_tris[:768].min()) + 1)] = (wide_verts, wide_tris.reshape(-1), fits)
    images = sr.scene.material_textures(16, seed=4)
    devices = {"lists": util.device_with_env(sr, SR_FILL_BLOCKS="1"), "none": util.device_with_env(sr, SR_FILL_BLOCKS="1", SR_NO_CHUNK_VERTS="1"),
               "lists, stored depth": util.device_with_env(sr, SR_FILL_BLOCKS="1", SR_STORE_DEPTH="1")}
    try:
        tex = {k: [sr.create_texture(d, im) for im in images] for k, d in devices.items()}
        tgt = {k: sr.RenderTarget(d, W, H, samples=samples, n_views=n_views) for k, d in devices.items()}
        cams = sr.scene.orbit_ubos(n_views + 1, W, H)
        for mesh_name, (v_, i_, has_lists) in meshes.items():
            mesh = {k: sr.create_mesh_from_data(d, v_, i_) for k, d in devices.items()}
            before = devices["lists"].draw_counts()["chunk_list_draws"]
            for k in range(2):
                u = cams[k:k + n_views]
                for name, d in devices.items():
                    sr.draw(d, tgt[name], mesh[name], u, tex[name])
                for v in range(n_views):
                    got = {name: {"tri": tgt[name].read_triangle_ids(v), "depth": tgt[name].read_depth(v), "bgra": tgt[name].read_color(v)} for name in devices}
                    if v % 4 == 0:
                        _check_view(got["lists"], oracle.draw(u[v], v_, i_, images, W, H, samples, threads=4))
                    for name in ("none", "lists, stored depth"):
                        for key in ("tri", "depth", "bgra"):
                            assert np.array_equal(got[name][key], got["lists"][key]), (mesh_name, k, v, name, key)
            used = devices["lists"].draw_counts()["chunk_list_draws"] - before
            assert (used > 0) == has_lists, (mesh_name, used)
            assert devices["none"].draw_counts()["chunk_list_draws"] == 0
            for m in mesh.values():
                m.destroy()
    finally:
        for d in devices.values():
            d.close()


@pytest.mark.parametrize("samples", [1, 4])
def test_metallic_and_roughness_maps_packed_into_one_chain(sr, oracle, samples, anisotropic):
    """The first draw that uses a metallic and a roughness map of the same size packs their red channels into one chain, and the shading
    pass fetches both with one set of loads.  The frame is the one a device without packing draws, bit for bit, and the oracle's: maps
    with different content in every channel, non-power-of-two maps, one map used for both, and -- after one of the pair has been
    destroyed and another map created in its place -- the new pair."""
    W, H = 320, 200
    verts, idx = sr.scene.uv_sphere(48, 36)
    ubo = sr.scene.default_camera_ubo(W, H)
    rng = np.random.default_rng(11)
    def noise(w, h):
        return rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    sets = [[noise(64, 64), noise(8, 8), noise(32, 32), noise(32, 32)], [noise(50, 20), noise(8, 8), noise(33, 7), noise(33, 7)]]
    packing, plain = util.device_with_env(sr), util.device_with_env(sr, SR_NO_PACKED_MAPS="1")
    try:
        for images in sets:
            got = {}
            for d in (packing, plain):
                got[d] = util.gpu_render(sr, d, verts, idx, images, ubo, W, H, samples)[0]
            _check_view(got[packing], oracle.draw(ubo, verts, idx, images, W, H, samples, threads=4))
            assert np.array_equal(got[packing]["bgra"], got[plain]["bgra"]) and np.array_equal(got[packing]["vis"], got[plain]["vis"])
        # one map for both; then the roughness map destroyed and replaced (the packed chain of the old pair must not be used for the new one)
        images = sets[0]
        mesh = sr.create_mesh_from_data(packing, verts, idx)
        tex = [sr.create_texture(packing, im) for im in images]
        tgt = sr.RenderTarget(packing, W, H, samples=samples)
        both = [tex[0], tex[1], tex[2], tex[2]]
        sr.draw(packing, tgt, mesh, ubo, both)
        _check_view({"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)},
                    oracle.draw(ubo, verts, idx, [images[0], images[1], images[2], images[2]], W, H, samples, threads=4))
        sr.draw(packing, tgt, mesh, ubo, tex)
        first = tgt.read_color(0)
        tex[3].destroy()
        other = noise(32, 32)
        tex[3] = sr.create_texture(packing, other)
        sr.draw(packing, tgt, mesh, ubo, tex)
        second = {"tri": tgt.read_triangle_ids(0), "depth": tgt.read_depth(0), "bgra": tgt.read_color(0)}
        _check_view(second, oracle.draw(ubo, verts, idx, [images[0], images[1], images[2], other], W, H, samples, threads=4))
        assert (second["bgra"] != first).any()
        tgt.destroy(); mesh.destroy()
        for t in tex:
            t.destroy()
    finally:
        packing.close(); plain.close()


def test_staged_resolve_gives_the_same_frames(sr, oracle, device):
    """A one-sample, one-view set-up launch of thousands of workgroups resolves its micro-triangles' samples through LDS (marked per lane,
    numbered per wave, resolved by whole waves); smaller launches resolve inside each lane's walk.  Forced on for small launches
    (SR_STAGED_MIN_BLOCKS=1), the frames are the default path's and the oracle's: sub-pixel triangles (a wave's samples fit one round),
    triangles of a few dozen pixels (several rounds per wave), a soup with coplanar duplicates, far-plane fragments outside the clip
    volume, a band of a frame."""
    staged = util.device_with_env(sr, SR_STAGED_MIN_BLOCKS="1")
    images = sr.scene.material_textures(64, seed=3)
    far_ubo, far_verts, far_idx = _window_triangles(sr, 96, 64, [[[3.0 + 5 * k, 2.0 + 3 * (k % 7), 1.0], [7.5 + 5 * k, 3.0 + 3 * (k % 7), 1.0], [4.0 + 5 * k, 8.0 + 3 * (k % 7), 1.0]] for k in range(17)])
    cases = [(sr.scene.uv_sphere(300, 200), sr.scene.default_camera_ubo(320, 200), 320, 200, None),
             (sr.scene.uv_sphere(64, 48), sr.scene.default_camera_ubo(640, 480), 640, 480, None),
             (sr.scene.uv_sphere(88, 88), sr.scene.default_camera_ubo(800, 600), 800, 600, (0, 200, 800, 150)),
             (util.random_soup(5000, 12, size=0.2), sr.scene.default_camera_ubo(400, 300), 400, 300, None),
             ((far_verts, far_idx), far_ubo, 96, 64, None)]
    try:
        for k, ((verts, idx), ubo, W, H, rect) in enumerate(cases):
            a = util.gpu_render(sr, staged, verts, idx, images, ubo, W, H, 1, rect=rect)[0]
            b = util.gpu_render(sr, device, verts, idx, images, ubo, W, H, 1, rect=rect)[0]
            assert np.array_equal(a["vis"], b["vis"]) and np.array_equal(a["bgra"], b["bgra"]), k
            assert a["stats"] == b["stats"], k
            assert a["stats"]["triangles_direct"] > 0, k
            _check_view(a, oracle.draw(ubo, verts, idx, images, W, H, 1, rect=rect, threads=4))
    finally:
        staged.close()


def test_non_power_of_two_and_mixed_size_textures(sr, oracle, device, anisotropic):
    rng = np.random.default_rng(2)
    images = [rng.integers(0, 256, size=(h, w, 4), dtype=np.uint8) for (w, h) in [(100, 37), (8, 8), (1, 64), (33, 5)]]
    verts, idx = sr.scene.uv_sphere(40, 30)
    verts[:, 6:8] *= 3.0   # exercise REPEAT
    ubo = sr.scene.default_camera_ubo(400, 300)
    got = util.gpu_render(sr, device, verts, idx, images, ubo, 400, 300, 1)[0]
    ref = oracle.draw(ubo, verts, idx, images, 400, 300, 1, threads=4)
    _check_view(got, ref)


@pytest.mark.parametrize("size", [(256, 256), (64, 16), (100, 37), (1, 64), (5, 1), (1, 1)])
def test_mip_chain_bit_exact(sr, oracle, device, size):
    w, h = size
    img = np.random.default_rng(w * 131 + h).integers(0, 256, size=(h, w, 4), dtype=np.uint8)
    tex = sr.create_texture(device, img)
    chain = oracle.build_mip_chain(img)
    assert tex.mip_levels == oracle.mip_levels(w, h)
    for level in range(tex.mip_levels):
        assert np.array_equal(tex.read_level(level), oracle.chain_level(chain, w, h, level)), f"level {level} differs"
    tex.destroy()


def test_golden_fixtures(sr, device):
    """The committed oracle outputs (tests/golden/make_golden.py) reproduced by the GPU path."""
    import make_golden
    for name in make_golden.SCENES:
        sc = make_golden.build_scene(name)
        g = np.load(make_golden.path_of(name))
        got = util.gpu_render(sr, device, sc["verts"], sc["idx"], sc["images"], sc["ubo"], sc["width"], sc["height"], sc["samples"])[0]
        _check_view(got, {"tri": g["tri"], "depth": g["depth"], "bgra": g["bgra"]})   # (the committed colours are trilinear, like the session's default sampler)


def test_cpp_host_app_matches_python_host(sr, device, tmp_path):
    """The C++ mirror of the reference's host API (vulkan-renderer_amd/host) drives the same library:
    its frame equals the Python host's up to the sphere's last-bit vertex differences."""
    import os
    import subprocess
    if os.environ.get("SR_TEST_EMULATED") == "1":
        pytest.skip("the C++ application links the GPU library; not part of the CPU emulation run")
    app = os.path.join(os.path.dirname(sr.LIB_PATH), "headless_app")
    assert os.path.exists(app), "build it with __graft_entry__.build()"
    out = tmp_path / "frame.ppm"
    subprocess.run([app, "320", "240", "48", "32", str(out)], check=True, timeout=120)
    data = out.read_bytes()
    header, rest = data.split(b"\n255\n", 1)
    assert header == b"P6\n320 240"
    rgb = np.frombuffer(rest, np.uint8).reshape(240, 320, 3)
    verts, idx = sr.scene.uv_sphere(48, 32)
    images = [sr.scene.albedo_pattern(2048), sr.scene.constant_rgba(64, 64, (128, 128, 255, 255)),
              sr.scene.gray_map(2048, 2048, 0, 254), sr.scene.gray_map(2048, 2048, 72, 255)]
    sr.DEFAULT_MAX_ANISOTROPY = None     # the application creates its textures with the library's default sampler (the reference's: maxAnisotropy 16)
    try:
        got = util.gpu_render(sr, device, verts, idx, images, sr.scene.default_camera_ubo(320, 240), 320, 240)[0]
    finally:
        sr.DEFAULT_MAX_ANISOTROPY = 1.0
    frac, worst = util.color_match_fraction(rgb, got["bgra"][..., [2, 1, 0]], 2)
    assert frac >= 0.995, (frac, worst)


def test_colour_bound_to_caller_memory(sr, oracle, device):
    """sr_target_bind_color: the bands of a frame land in caller-owned device memory (what the sort-first composite
    hands to RCCL) and composite to the single-target frame bit for bit.  The memory comes from the HIP runtime the
    library itself uses; when torch sees the GPU too, the same is done once more into torch tensors."""
    import ctypes
    import os
    if os.environ.get("SR_TEST_EMULATED") == "1":
        pytest.skip("needs caller-owned GPU memory; not part of the CPU emulation run")
    from softraster import distributed as d
    hip = ctypes.CDLL("libamdhip64.so.7")   # already mapped by the library: the same runtime, the same context
    hip.hipMalloc.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t]
    hip.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    hip.hipFree.argtypes = [ctypes.c_void_p]
    W, H = 256, 192
    verts, idx = sr.scene.uv_sphere(32, 24)
    images = sr.scene.material_textures(64)
    ubo = sr.scene.default_camera_ubo(W, H)
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = [sr.create_texture(device, im) for im in images]
    bands = d.split_bands(W, H, 3)
    try:
        frame_mem = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(frame_mem), W * H * 4) == 0
        for (x, y, w, h) in bands:       # every band renders straight into its rows of the one frame
            tgt = sr.RenderTarget(device, W, H, rect=(x, y, w, h))
            tgt.bind_color(frame_mem.value + y * W * 4)
            sr.draw(device, tgt, mesh, ubo, tex)
            device.synchronize()
            tgt.destroy()
        frame = np.empty((H, W, 4), np.uint8)
        assert hip.hipMemcpy(frame.ctypes.data, frame_mem, frame.nbytes, 2) == 0     # hipMemcpyDeviceToHost
        # a draw stores only the pixels whose colour changes; memory the caller has written into is bound again, and the
        # next draw then stores every pixel: scribble over the frame, bind, draw twice (the second stores nothing new)
        tgt = sr.RenderTarget(device, W, H)
        tgt.bind_color(frame_mem.value)
        sr.draw(device, tgt, mesh, ubo, tex)
        device.synchronize()
        scribble = np.full((H, W, 4), 0x5A, np.uint8)
        assert hip.hipMemcpy(frame_mem, scribble.ctypes.data, scribble.nbytes, 1) == 0   # hipMemcpyHostToDevice
        tgt.bind_color(frame_mem.value)
        for _ in range(2):
            sr.draw(device, tgt, mesh, ubo, tex)
            device.synchronize()
            again = np.empty((H, W, 4), np.uint8)
            assert hip.hipMemcpy(again.ctypes.data, frame_mem, again.nbytes, 2) == 0
            assert np.array_equal(again, frame)
        tgt.destroy()
        assert hip.hipFree(frame_mem) == 0
        ref = oracle.draw(ubo, verts, idx, images, W, H)
        frac, worst = util.color_match_fraction(frame, ref["bgra"], 1)
        assert frac >= 0.999, (frac, worst)
        full = util.gpu_render(sr, device, verts, idx, images, ubo, W, H, textures=tex)[0]
        assert np.array_equal(frame, full["bgra"])       # bands composite to the single-target frame bit for bit
        import torch
        if torch.cuda.is_available():
            parts = []
            for band in bands:
                t, tgt = d.render_band_to_tensor(sr, device, mesh, tex, ubo, W, H, band)
                parts.append(t.cpu().numpy())
                tgt.destroy()
            assert np.array_equal(np.concatenate(parts, axis=0), full["bgra"])
    finally:
        mesh.destroy()
        for t in tex:
            t.destroy()
