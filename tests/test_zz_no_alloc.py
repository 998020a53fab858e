"""sr_draw_no_alloc (include/softraster.h): the draw for callers that cannot allocate keeps every intermediate buffer in a
static array of the library's device image.  Its frames must equal sr_draw's bit for bit -- visibility words and colour
-- on two calls in a row, the second with a smaller input, which finds the first call's data still in the array; limits
are reported, not overrun.  Collected last on purpose (file name): a failure here hides no other test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(sr, device, verts, idx, images, ubos, W, H, samples):
    """The same draw through sr_draw and through sr_draw_no_alloc, each into a target of its own."""
    ubos = np.atleast_2d(ubos)
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = [sr.create_texture(device, im) for im in images]
    targets = [sr.RenderTarget(device, W, H, samples=samples, n_views=ubos.shape[0]) for _ in range(2)]
    try:
        out = []
        for tgt, no_alloc in zip(targets, (False, True)):
            for _ in range(2):     # twice: the second draw of a target takes the paths that reuse what the first left
                sr.draw(device, tgt, mesh, ubos, tex, no_alloc=no_alloc)
            out.append(([tgt.read_visibility(v) for v in range(ubos.shape[0])], [tgt.read_color(v) for v in range(ubos.shape[0])], tgt.stats()))
        return out
    finally:
        for t in targets:
            t.destroy()
        mesh.destroy()
        for t in tex:
            t.destroy()


def test_same_frames_as_sr_draw_on_two_calls_in_a_row(sr, oracle, device):
    import util
    cases = [
        # a large input first: 120 k sub-pixel and small triangles, three views, 4 samples -- tile lists, micro path
        (sr.scene.uv_sphere(300, 200), sr.scene.material_textures(128), sr.scene.orbit_ubos(3, 640, 360), 640, 360, 4),
        # then a smaller one, which finds the first call's lists and counters still in the array: a soup with coplanar
        # duplicates and a quad that reaches behind the eye (the clipper), one view, one sample
        (util.random_soup(700, 3, size=0.9), sr.scene.material_textures(32, seed=2), sr.scene.default_camera_ubo(200, 120), 200, 120, 1),
    ]
    quad_v, quad_i = sr.scene.screen_quad(z=0.0, half=500.0)
    sv, si = cases[1][0]
    cases[1] = ((np.concatenate([sv, quad_v]), np.concatenate([si, quad_i + sv.shape[0]]).astype(np.uint32)),) + cases[1][1:]
    for (verts, idx), images, ubos, W, H, samples in cases:
        (vis_a, col_a, st_a), (vis_b, col_b, st_b) = _both(sr, device, verts, idx, images, ubos, W, H, samples)
        for v in range(len(vis_a)):
            assert np.array_equal(vis_a[v], vis_b[v]), f"view {v}: visibility words differ at {int((vis_a[v] != vis_b[v]).sum())} samples"
            assert np.array_equal(col_a[v], col_b[v]), f"view {v}: colour differs"
            assert (vis_a[v] & 0xFFFFFFFF != sr.NO_TRIANGLE).any()
        assert {k: x for k, x in st_a.items() if k not in ("tile_refs", "passes")} == {k: x for k, x in st_b.items() if k not in ("tile_refs", "passes")}
    assert st_b["triangles_clipped"] == 2      # the second case went through the clipper
    # and against the oracle, so that "equal" is not "equally wrong"
    (verts, idx), images, ubo, W, H, samples = cases[1]
    ref = oracle.draw(ubo, verts, idx, images, W, H, samples)
    assert np.array_equal((vis_b[0] & 0xFFFFFFFF).astype(np.uint32), ref["tri"]) and np.array_equal((vis_b[0] >> 32).astype(np.uint32), ref["depth"].view(np.uint32))


def test_limits_are_reported(sr, device):
    assert sr.load_library().sr_no_alloc_scratch_bytes() == 64 << 20
    images = sr.scene.material_textures(16)
    tex = [sr.create_texture(device, im) for im in images]
    try:
        # more views than travel as kernel arguments
        verts, idx = sr.scene.uv_sphere(8, 8)
        mesh = sr.create_mesh_from_data(device, verts, idx)
        tgt = sr.RenderTarget(device, 64, 48, n_views=9)
        with pytest.raises(sr.SoftRasterError) as e:
            sr.draw(device, tgt, mesh, sr.scene.orbit_ubos(9, 64, 48), tex, no_alloc=True)
        assert e.value.status == -5 and "at most 8 views" in str(e.value)
        sr.draw(device, tgt, mesh, sr.scene.orbit_ubos(8, 64, 48), tex, no_alloc=True)      # eight are fine
        assert (tgt.read_triangle_ids(7) != sr.NO_TRIANGLE).any()
        tgt.destroy(); mesh.destroy()
        # a mesh whose scratch does not fit the array (2.2 M triangles: 48 bytes each)
        verts, idx = sr.scene.uv_sphere(1100, 1000)
        mesh = sr.create_mesh_from_data(device, verts, idx)
        tgt = sr.RenderTarget(device, 320, 200)
        with pytest.raises(sr.SoftRasterError) as e:
            sr.draw(device, tgt, mesh, sr.scene.default_camera_ubo(320, 200), tex, no_alloc=True)
        assert e.value.status == -5 and "bytes of scratch" in str(e.value)
        sr.draw(device, tgt, mesh, sr.scene.default_camera_ubo(320, 200), tex)              # the ordinary entry point takes it
        assert (tgt.read_triangle_ids(0) != sr.NO_TRIANGLE).any()
        tgt.destroy(); mesh.destroy()
        # more triangles through the eye plane than this entry point's clip list holds (1024): reported when the target is read,
        # not handed over with triangles missing; the ordinary entry point (65536 entries) draws the frame
        import util
        rng = np.random.default_rng(5)
        verts = np.zeros((3 * 1800, 8), np.float32)
        verts[:, :3] = rng.uniform(-2.0, 2.0, size=(3 * 1800, 3))
        verts[:, 5] = 1.0
        idx = np.arange(3 * 1800, dtype=np.uint32)
        eye = np.array([0.0, 0.0, 0.0])
        ubo = sr.scene.make_ubo(np.eye(4, dtype=np.float32), sr.scene.look_at(eye, (0.0, 1.0, 0.0), (0.0, 0.0, 1.0)), sr.scene.perspective(0.8, 1.6, 0.1, 10.0), eye)
        mesh = sr.create_mesh_from_data(device, verts, idx)
        tgt = sr.RenderTarget(device, 160, 100)
        sr.draw(device, tgt, mesh, ubo, tex, no_alloc=True)
        with pytest.raises(sr.SoftRasterError) as e:
            tgt.read_triangle_ids(0)
        assert e.value.status == -5 and "clip list" in str(e.value)
        sr.draw(device, tgt, mesh, ubo, tex)
        assert tgt.stats()["triangles_clipped"] > 1024 and tgt.stats()["triangles_unprojectable"] == 0
        tgt.destroy(); mesh.destroy()
    finally:
        for t in tex:
            t.destroy()
