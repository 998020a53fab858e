"""Pins the CPU oracle with hand-derivable answers.  The reference has no tests or golden images for
the draw path ("parity unpinned"), so these micro-scenes are what stands behind the oracle: camera
matrices, coverage and fill rule, face culling, depth test, multisample positions, clip handling."""
import json
import math
import os

import numpy as np
import pytest

import util
from softraster import scene

GOLD = json.load(open(os.path.join(util.GOLDEN_DIR, "camera_known_answers.json")))


def ortho_ubo():
    """Identity model/view and a projection that passes x, y, z through with w = 1: NDC = position."""
    eye = np.eye(4, dtype=np.float32)
    return scene.make_ubo(eye, eye, eye, (0.0, 0.0, 5.0))


def px_to_ndc(px, py, W, H):
    """Window position (pixels, y down) -> NDC under the reference's flipped viewport."""
    return 2.0 * px / W - 1.0, 1.0 - 2.0 * py / H


def rect_mesh(x0, y0, x1, y1, W, H, z=0.5, flip=False, diag_other=False):
    """Axis-aligned rectangle given in window pixels, two triangles, counter-clockwise on screen."""
    ax, ay = px_to_ndc(x0, y0, W, H); bx, by = px_to_ndc(x1, y1, W, H)
    # on screen (y down): TL=(x0,y0) BL=(x0,y1) BR=(x1,y1) TR=(x1,y0); CCW as displayed = TL, BL, BR, TR
    v = np.zeros((4, 8), np.float32)
    v[:, 0:3] = [(ax, ay, z), (ax, by, z), (bx, by, z), (bx, ay, z)]
    v[:, 5] = 1.0
    v[:, 6:8] = [(0, 0), (0, 1), (1, 1), (1, 0)]
    idx = np.array([0, 1, 2, 0, 2, 3] if not diag_other else [0, 1, 3, 1, 2, 3], np.uint32)
    if flip:
        idx = idx.reshape(-1, 3)[:, ::-1].reshape(-1).copy()
    return v, idx


TEX = [scene.constant_rgba(2, 2, (200, 100, 50, 255)), None, scene.constant_rgba(2, 2, (0, 0, 0, 255)), scene.constant_rgba(2, 2, (180, 180, 180, 255))]
TEX[1] = TEX[0]


def test_camera_known_answers(oracle):
    ubo = scene.default_camera_ubo(800, 600)
    view = ubo[16:32].reshape(4, 4).T
    proj = ubo[32:48].reshape(4, 4).T
    assert np.allclose(view, np.array(GOLD["view_rows"], np.float32), atol=2e-7)
    p = GOLD["proj_800x600"]
    assert proj[0, 0] == pytest.approx(p["P00"], abs=2e-7) and proj[1, 1] == pytest.approx(p["P11"], abs=3e-7)
    assert proj[2, 2] == pytest.approx(p["P22"], abs=2e-7) and proj[2, 3] == pytest.approx(p["P23"], abs=2e-8) and proj[3, 2] == p["P32"]
    assert scene.default_camera_ubo(1920, 1080)[32] == pytest.approx(GOLD["proj_16x9_P00"], abs=2e-7)
    # the oracle's own GLM restatement agrees with the host-side one
    v = oracle.look_at((2, 2, 2), (0, 0, 0), (0, 0, 1))
    assert np.allclose(v.reshape(4, 4).T, view, atol=2e-7)
    pr = oracle.perspective(math.radians(45.0), 800 / 600, 0.1, 10.0)
    assert np.allclose(pr.reshape(4, 4).T, proj, atol=3e-7)
    # origin through the whole vertex stage
    verts = np.zeros((1, 8), np.float32)
    clip, screen = oracle.vertex_stage(ubo, verts, 800, 600)
    assert clip[0, 2] / clip[0, 3] == pytest.approx(GOLD["origin_ndc_z"], abs=1e-6)
    assert screen["z"][0] == pytest.approx(GOLD["origin_ndc_z"], abs=1e-6)
    assert screen["X"][0] == 400 * 256 and screen["Y"][0] == 300 * 256   # centre of an 800x600 frame, 1/256 pixel
    assert ubo.nbytes == GOLD["ubo_bytes"]


def test_quad_on_pixel_boundaries_covers_every_pixel_once(oracle):
    W, H = 64, 32
    for diag_other in (False, True):
        v, i = rect_mesh(8, 4, 40, 20, W, H, diag_other=diag_other)
        r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)
        tri = r["tri"][..., 0]
        inside = np.zeros((H, W), bool); inside[4:20, 8:40] = True
        assert np.array_equal(tri != 0xFFFFFFFF, inside)
        assert set(np.unique(tri[inside])) == {0, 1}       # shared diagonal: no hole, no double hit
        assert r["stats"]["samples_passed"] == 32 * 16      # each sample written exactly once
        assert (r["depth"][..., 0][inside] == np.float32(0.5)).all()
        assert (r["depth"][..., 0][~inside] == 1.0).all()
        assert (r["bgra"][~inside] == np.array([0, 0, 0, 255], np.uint8)).all()   # clear colour (0,0,0,1)


def test_back_faces_are_culled(oracle):
    W, H = 32, 32
    v, i = rect_mesh(4, 4, 28, 28, W, H, flip=True)       # clockwise on the displayed image
    r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)
    assert (r["tri"] == 0xFFFFFFFF).all() and r["stats"]["tris_front"] == 0


def test_top_left_fill_rule(oracle):
    """Edges through pixel centres: left and top edges own the centre, right and bottom ones do not."""
    W, H = 16, 16
    v, i = rect_mesh(2.5, 1.5, 6.5, 4.5, W, H)
    r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)
    cov = r["tri"][..., 0] != 0xFFFFFFFF
    expect = np.zeros((H, W), bool); expect[1:4, 2:6] = True
    assert np.array_equal(cov, expect)


def test_depth_less_and_first_primitive_wins_ties(oracle):
    W, H = 32, 32
    near, i0 = rect_mesh(4, 4, 20, 20, W, H, z=0.25)
    far, i1 = rect_mesh(10, 10, 28, 28, W, H, z=0.75)
    for order in ((near, far), (far, near)):
        v = np.concatenate(order); i = np.concatenate([i0, i1 + 4]).astype(np.uint32)
        r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)
        d = r["depth"][..., 0]
        assert (d[10:20, 10:20] == np.float32(0.25)).all()          # overlap: nearer wins whatever the order
        assert (np.abs(d[20:28, 20:28] - np.float32(0.75)) <= 6e-8).all()   # l0+l1+l2 is 1 only to an ulp (documented convention)
    # coplanar duplicates: LESS keeps the first primitive, i.e. the lowest triangle ids
    v = np.concatenate([near, near]); i = np.concatenate([i0, i0 + 4]).astype(np.uint32)
    r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)
    assert set(np.unique(r["tri"][4:20, 4:20, 0])) == {0, 1}


def test_depth_outside_clip_volume_is_dropped(oracle):
    W, H = 16, 16
    for z in (-0.25, 1.25):
        v, i = rect_mesh(2, 2, 14, 14, W, H, z=z)
        r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)
        assert (r["tri"] == 0xFFFFFFFF).all()
    v, i = rect_mesh(2, 2, 14, 14, W, H, z=1.0)           # depth 1.0 is not LESS than the cleared 1.0
    r = oracle.draw(ortho_ubo(), v, i, TEX, W, H)           # (weights sum to 1 only to an ulp: a sample may land just below)
    assert (r["depth"][r["tri"] != 0xFFFFFFFF] < 1.0).all() and (r["depth"][r["tri"] == 0xFFFFFFFF] == 1.0).all()
    v, i = rect_mesh(2, 2, 14, 14, W, H, z=0.0)
    assert (oracle.draw(ortho_ubo(), v, i, TEX, W, H)["tri"][2:14, 2:14] != 0xFFFFFFFF).all()


def test_standard_4x_sample_positions(oracle):
    """Right edge at x = 4.5: in pixel column 4 only the samples left of 0.5 are covered
    (standard locations: s0 x=.375, s1 x=.875, s2 x=.125, s3 x=.625)."""
    W, H = 16, 8
    v, i = rect_mesh(0, 0, 4.5, 8, W, H)
    r = oracle.draw(ortho_ubo(), v, i, TEX, W, H, samples=4)
    cov = r["tri"] != 0xFFFFFFFF
    assert cov[:, :4, :].all() and not cov[:, 5:, :].any()
    assert (cov[:, 4, :] == np.array([True, False, True, False])).all()
    # bottom edge at y = 2.25: s0 (y=.125) covered in row 2, others (y=.375,.625,.875) not
    v, i = rect_mesh(0, 0, 16, 2.25, W, H)
    cov = oracle.draw(ortho_ubo(), v, i, TEX, W, H, samples=4)["tri"] != 0xFFFFFFFF
    assert (cov[2, :, :] == np.array([True, False, False, False])).all()
    # the resolve averages the stored 8-bit samples in linear light: 2 of 4 samples of a pixel covered
    v, i = rect_mesh(0, 0, 4.5, 8, W, H)
    r = oracle.draw(ortho_ubo(), v, i, TEX, W, H, samples=4)
    dec, _ = oracle.srgb_tables()
    full = r["bgra"][3, 2].astype(int); half = r["bgra"][3, 4].astype(int)
    for c in range(3):
        assert half[c] == oracle.srgb_encode8((2 * dec[full[c]] + 2 * dec[0]) / 4)
    assert half[3] == 255


def test_unprojectable_triangles_without_the_clipper_are_dropped_and_counted(oracle):
    W, H = 32, 32
    ubo = scene.default_camera_ubo(W, H)
    v, i = scene.screen_quad(z=0.0, half=500.0)           # spans behind the eye: needs clipping
    oracle.set_near_clip(False)
    try:
        r = oracle.draw(ubo, v, i, TEX, W, H)
    finally:
        oracle.set_near_clip(True)    # what the session's other tests expect
    assert r["stats"]["tris_clip_culled"] == 2 and (r["tri"] == 0xFFFFFFFF).all()
    behind, i = scene.screen_quad(z=0.0, half=1.0)
    behind[:, 0:3] += (10, 10, 10)                          # entirely behind the eye: silently invisible
    r = oracle.draw(ubo, behind, i, TEX, W, H)
    assert r["stats"]["tris_clip_culled"] == 0 and (r["tri"] == 0xFFFFFFFF).all()


@pytest.mark.parametrize("half", [50.0, 5000.0])
def test_near_plane_clipper_against_a_ray_cast(oracle, half):
    """sro_set_near_clip: a ground plane that passes under and behind the camera (two triangles with a vertex behind the
    eye; at +-5000 units their cuts at the near plane also leave the guard band and the side planes come in).  The
    covered pixels are exactly those whose centre ray hits the plane inside the quad between the near and the far
    plane, and the depth is the plane's; a quad cut short in front of the camera (one of its triangles needs no
    clipping) gets the same colours."""
    W, H = 320, 200
    eye = np.array([0.3, -2.0, 1.0])

    def quad(y0):
        v = np.zeros((4, 8), np.float32)
        v[:, :3] = [[-half, y0, 0], [half, y0, 0], [half, half, 0], [-half, half, 0]]
        v[:, 5] = 1.0
        v[:, 6:8] = v[:, :2] * 0.1
        return v
    idx = np.array([0, 1, 2, 0, 2, 3], np.uint32)
    view, proj = scene.look_at(eye, (0.0, 3.0, 0.0), (0, 0, 1)), scene.perspective(math.radians(45.0), W / H, 0.1, 10.0)
    ubo = scene.make_ubo(np.eye(4, dtype=np.float32), view, proj, eye)
    tex = scene.material_textures(64, seed=1)
    oracle.set_near_clip(False)
    off = oracle.draw(ubo, quad(-half), idx, tex, W, H, 1, shade=False)
    assert off["stats"]["tris_clip_culled"] == 2 and (off["tri"] == 0xFFFFFFFF).all()
    oracle.set_near_clip(True)
    try:
        r = oracle.draw(ubo, quad(-half), idx, tex, W, H, 1, threads=4)
        short = oracle.draw(ubo, quad(-1.0), idx, tex, W, H, 1, threads=4)
    finally:
        oracle.set_near_clip(True)   # what the session's other tests expect
    assert r["stats"]["tris_clipped"] == 2 and r["stats"]["tris_clip_culled"] == 0
    # the same frame by ray casting, in double precision
    pv = proj.astype(np.float64) @ view.astype(np.float64)
    inv = np.linalg.inv(pv)
    ys, xs = np.mgrid[0:H, 0:W]
    nx, ny = (xs + 0.5 - W / 2) / (W / 2), (H / 2 - (ys + 0.5)) / (H / 2)

    def unproject(z):
        p = np.stack([nx, ny, np.full_like(nx, z), np.ones_like(nx)], -1) @ inv.T
        return p[..., :3] / p[..., 3:4]
    a, b = unproject(0.0), unproject(1.0)
    t = a[..., 2] / (a[..., 2] - b[..., 2])
    hit = a + t[..., None] * (b - a)
    expected = (t >= 0) & (t <= 1) & (np.abs(hit[..., 0]) <= half) & (np.abs(hit[..., 1]) <= half)
    got = r["tri"][..., 0] != 0xFFFFFFFF
    assert expected.sum() > W * H // 2 and np.array_equal(got, expected)
    clip = np.concatenate([hit, np.ones_like(hit[..., :1])], -1) @ pv.T
    assert np.abs(r["depth"][..., 0][got] - (clip[..., 2] / clip[..., 3])[got]).max() < (1e-5 if half < 100 else 1e-4)
    both = got & (short["tri"][..., 0] != 0xFFFFFFFF)
    assert both.sum() > W * H // 2
    assert np.abs(r["bgra"].astype(int) - short["bgra"].astype(int)).max(axis=-1)[both].max() <= 1


def test_sphere_is_front_facing_from_outside(oracle):
    """The synthetic mesh must be counter-clockwise seen from outside: the hemisphere facing the
    default camera is drawn, and depth at the centre is the near pole of the unit sphere."""
    W, H = 200, 150
    v, i = scene.uv_sphere(32, 32)
    assert v.shape == (33 * 33, 8) and i.size == 3 * 2 * 32 * 32
    r = oracle.draw(scene.default_camera_ubo(W, H), v, i, TEX, W, H)
    assert r["tri"][H // 2, W // 2, 0] != 0xFFFFFFFF
    # eye distance sqrt(12), nearest point at distance sqrt(12)-1: NDC z = (f/(n-f))*(-d) ... / d
    d = math.sqrt(12.0) - 1.0
    zn, zf = 0.1, 10.0
    expect = (zf / (zn - zf) * -d - zf * zn / (zf - zn)) / d
    assert r["depth"][H // 2, W // 2, 0] == pytest.approx(expect, abs=2e-4)
    assert 0.3 < (r["tri"] != 0xFFFFFFFF).mean() < 0.5


def test_rect_target_is_a_crop(oracle):
    W, H = 96, 64
    v, i = scene.uv_sphere(24, 16)
    ubo = scene.default_camera_ubo(W, H)
    tex = scene.material_textures(32)
    full = oracle.draw(ubo, v, i, tex, W, H, samples=4)
    part = oracle.draw(ubo, v, i, tex, W, H, samples=4, rect=(17, 9, 50, 40))
    for k in ("tri", "depth", "bgra"):
        assert np.array_equal(part[k], full[k][9:49, 17:67])


def test_threads_do_not_change_the_result(oracle):
    W, H = 120, 90
    v, i = util.random_soup(400, seed=3, size=0.9)
    ubo = scene.default_camera_ubo(W, H)
    tex = scene.material_textures(16, seed=2)
    a = oracle.draw(ubo, v, i, tex, W, H, samples=4, threads=1)
    b = oracle.draw(ubo, v, i, tex, W, H, samples=4, threads=7)
    for k in ("tri", "depth", "bgra"):
        assert np.array_equal(a[k], b[k])


def test_oracle_reproduces_committed_goldens(oracle):
    import make_golden
    for name in make_golden.SCENES:
        sc = make_golden.build_scene(name)
        g = np.load(make_golden.path_of(name))
        r = oracle.draw(sc["ubo"], sc["verts"], sc["idx"], sc["images"], sc["width"], sc["height"], sc["samples"])
        assert np.array_equal(r["tri"], g["tri"]) and np.array_equal(r["depth"].view(np.uint32), g["depth"].view(np.uint32)), name
        frac, worst = util.color_match_fraction(r["bgra"], g["bgra"], 0)
        assert frac == 1.0, (name, frac, worst)
