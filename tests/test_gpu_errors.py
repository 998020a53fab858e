"""Error behaviour at the C ABI: status codes instead of aborts, nothing undefined reaches the GPU."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_index_out_of_range_is_rejected(sr, device):
    verts, idx = sr.scene.screen_quad()
    idx = idx.copy(); idx[4] = 4
    with pytest.raises(sr.SoftRasterError) as e:
        sr.create_mesh_from_data(device, verts, idx)
    assert e.value.status == -4


def test_bad_arguments(sr, device):
    with pytest.raises(sr.SoftRasterError):
        sr.RenderTarget(device, 0, 10)
    with pytest.raises(sr.SoftRasterError):
        sr.RenderTarget(device, 64, 64, samples=2)
    with pytest.raises(sr.SoftRasterError):
        sr.RenderTarget(device, 64, 64, rect=(32, 32, 64, 8))
    with pytest.raises(sr.SoftRasterError):
        sr.create_texture(device, np.zeros((0, 0, 4), np.uint8), 0, 0)
    tgt = sr.RenderTarget(device, 64, 64)
    with pytest.raises(sr.SoftRasterError):
        tgt.read_color()          # nothing drawn yet
    verts, idx = sr.scene.screen_quad()
    mesh = sr.create_mesh_from_data(device, verts, idx)
    tex = sr.create_texture(device, np.full((4, 4, 4), 255, np.uint8))
    with pytest.raises(sr.SoftRasterError):
        sr.draw(device, tgt, mesh, sr.scene.default_camera_ubo(64, 64), [tex, None, None, tex])   # metallic missing
    with pytest.raises(sr.SoftRasterError):
        sr.draw(device, tgt, mesh, sr.scene.orbit_ubos(2, 64, 64), [tex, None, tex, tex])          # more views than layers
    tex.destroy(); mesh.destroy(); tgt.destroy()


def test_wrong_device_ordinal(sr):
    with pytest.raises(sr.SoftRasterError) as e:
        sr.Device(4096)
    assert e.value.status == -6


def test_trailing_indices_are_ignored(sr, oracle, device):
    import util
    verts, idx = sr.scene.screen_quad()
    idx5 = idx[:5]
    ubo = sr.scene.default_camera_ubo(96, 64)
    images = sr.scene.material_textures(16)
    got = util.gpu_render(sr, device, verts, idx5, images, ubo, 96, 64)[0]
    ref = oracle.draw(ubo, verts, idx[:3], images, 96, 64)
    util.assert_visibility_equal(got, ref)
