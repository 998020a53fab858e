"""Python host binding of the gfx950 rasteriser's C ABI (include/softraster.h) via ctypes.

Mirrors the reference's host API for the draw path: ``Device`` stands for ``vkh::GPUDevice``
(src/vulkan_helper/gpu_device.hpp:23-110), ``create_mesh_from_data`` for src/mesh.hpp:69-72,
``create_texture`` / ``destroy_texture`` for src/texture.hpp:17-24, ``RenderTarget`` for the
swap-chain image plus multisampled colour/depth attachments (src/main.cpp:723-738) and
``draw`` for update_uniform_buffer + the recorded draw + submit (src/main.cpp:988-1004, 897-959).

There is no CPU fallback: loading fails loudly if the HIP library is missing, and creating a
device fails if no gfx950 GPU is visible.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import scene  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "lib", "libsoftraster.so")

READ_COLOR_BGRA8, READ_DEPTH_F32, READ_TRI_ID_U32, READ_VIS_U64 = 0, 1, 2, 3
NO_TRIANGLE = 0xFFFFFFFF

EXPORTED_SYMBOLS = (
    "sr_device_create", "sr_device_destroy", "sr_device_synchronize", "sr_status_string", "sr_last_error_string",
    "sr_mesh_create", "sr_mesh_destroy", "sr_sampler_default", "sr_texture_create_rgba8_srgb", "sr_texture_destroy",
    "sr_texture_mip_levels", "sr_texture_read_level", "sr_target_create", "sr_target_destroy", "sr_target_bind_color",
    "sr_target_readback", "sr_draw", "sr_draw_get_stats", "sr_profile_enable", "sr_profile_get", "sr_debug_vertex_stage",
    "sr_texture_size", "sr_texture_create_from_file", "sr_image_read_file", "sr_image_free",
    "sr_model_load", "sr_model_mesh_count", "sr_model_mesh", "sr_model_destroy", "sr_model_read_file", "sr_model_mesh_data",
    "sr_draw_no_alloc", "sr_no_alloc_scratch_bytes", "sr_device_set_normal_mapping", "sr_device_get_draw_counts",
)


class SoftRasterError(RuntimeError):
    def __init__(self, status: int, what: str, detail: str):
        super().__init__(f"{what}: status {status} ({detail})")
        self.status = status


class SamplerDesc(C.Structure):
    _fields_ = [("min_lod", C.c_float), ("max_lod", C.c_float), ("mip_lod_bias", C.c_float), ("max_anisotropy", C.c_float)]


class Rect(C.Structure):
    _fields_ = [("x", C.c_int32), ("y", C.c_int32), ("width", C.c_int32), ("height", C.c_int32)]


class DrawStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("vertices", "triangles", "triangles_binned", "tile_refs", "triangles_unprojectable",
                                           "shaded_pixels", "visible_triangles", "pixels", "triangles_direct", "triangles_clipped", "passes")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class DrawCounts(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("draws", "lean_draws", "direct_draws", "lean_redraws", "overflow_redraws", "chunks_tested", "chunks_culled", "chunk_list_draws")]


class StageTimes(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("clear_ms", "vertex_ms", "setup_ms", "bin_ms", "scan_ms", "fill_ms", "raster_ms", "fragment_ms", "total_ms")]

    def as_dict(self):
        return {n: float(getattr(self, n)) for n, _ in self._fields_}


_lib = None

# maxAnisotropy of the sampler create_texture / create_texture_from_file make when none is passed.  None: the library's
# own default, which is the reference's (16, src/texture.cpp:255-256).  A process that wants plain trilinear filtering
# everywhere sets it to 1.0.
DEFAULT_MAX_ANISOTROPY = None


def default_sampler(max_anisotropy: float | None = None) -> "SamplerDesc":
    """The sampler of src/texture.cpp:246-263, optionally with another maxAnisotropy (<= 1: isotropic)."""
    s = SamplerDesc()
    load_library().sr_sampler_default(C.byref(s))
    if max_anisotropy is not None:
        s.max_anisotropy = float(max_anisotropy)
    return s


def _sampler_arg(sampler):
    if sampler is None and DEFAULT_MAX_ANISOTROPY is not None:
        sampler = default_sampler(DEFAULT_MAX_ANISOTROPY)
    return C.byref(sampler) if sampler is not None else None


def load_library():
    """Loads lib/libsoftraster.so (built by __graft_entry__.build()).  Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'`; "
                                "there is no CPU fallback for the draw path")
    L = C.CDLL(LIB_PATH)
    vp, u32, sz = C.c_void_p, C.c_uint32, C.c_size_t
    L.sr_status_string.restype = C.c_char_p
    L.sr_status_string.argtypes = [C.c_int]
    L.sr_last_error_string.restype = C.c_char_p
    L.sr_device_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.sr_device_destroy.argtypes = [vp]
    L.sr_device_synchronize.argtypes = [vp]
    L.sr_mesh_create.argtypes = [vp, vp, u32, vp, u32, C.POINTER(vp)]
    L.sr_mesh_destroy.argtypes = [vp, vp]
    L.sr_sampler_default.argtypes = [C.POINTER(SamplerDesc)]
    L.sr_sampler_default.restype = None
    L.sr_texture_create_rgba8_srgb.argtypes = [vp, vp, C.c_int, C.c_int, C.POINTER(SamplerDesc), C.POINTER(vp)]
    L.sr_texture_destroy.argtypes = [vp, vp]
    L.sr_texture_mip_levels.argtypes = [vp]
    L.sr_texture_read_level.argtypes = [vp, vp, C.c_int, vp, sz]
    L.sr_target_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, u32, C.POINTER(Rect), C.POINTER(vp)]
    L.sr_target_destroy.argtypes = [vp, vp]
    L.sr_target_bind_color.argtypes = [vp, vp, vp]
    L.sr_target_readback.argtypes = [vp, vp, C.c_int, u32, vp, sz]
    L.sr_draw.argtypes = [vp, vp, vp, vp, u32, sz, C.POINTER(vp)]
    L.sr_draw_no_alloc.argtypes = [vp, vp, vp, vp, u32, sz, C.POINTER(vp)]
    L.sr_no_alloc_scratch_bytes.restype = sz
    L.sr_device_set_normal_mapping.argtypes = [vp, C.c_int]
    L.sr_draw_get_stats.argtypes = [vp, vp, C.POINTER(DrawStats)]
    L.sr_device_get_draw_counts.argtypes = [vp, C.POINTER(DrawCounts)]
    L.sr_profile_enable.argtypes = [vp, C.c_int]
    L.sr_profile_get.argtypes = [vp, C.POINTER(StageTimes)]
    L.sr_debug_vertex_stage.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
    L.sr_texture_size.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sr_texture_create_from_file.argtypes = [vp, C.c_char_p, C.POINTER(SamplerDesc), C.POINTER(vp)]
    L.sr_image_read_file.argtypes = [C.c_char_p, C.POINTER(vp), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.sr_image_free.argtypes = [vp]
    L.sr_image_free.restype = None
    L.sr_model_load.argtypes = [vp, C.c_char_p, C.POINTER(vp)]
    L.sr_model_read_file.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.sr_model_mesh_count.argtypes = [vp]
    L.sr_model_mesh_count.restype = u32
    L.sr_model_mesh.argtypes = [vp, u32]
    L.sr_model_mesh.restype = vp
    L.sr_model_destroy.argtypes = [vp, vp]
    L.sr_model_mesh_data.argtypes = [vp, u32, C.POINTER(vp), C.POINTER(u32), C.POINTER(vp), C.POINTER(u32)]
    _lib = L
    return L


def _check(status: int, what: str):
    if status != 0:
        L = load_library()
        detail = f"{L.sr_status_string(status).decode()}; {L.sr_last_error_string().decode()}"
        raise SoftRasterError(status, what, detail)


SCREEN_DTYPE = np.dtype([("X", "<i4"), ("Y", "<i4"), ("z", "<f4"), ("rw", "<f4")])


class Device:
    """One GPU and its stream.  Not thread-safe, like the reference's single-threaded host."""

    def __init__(self, hip_ordinal: int = 0):
        self._lib = load_library()
        self._h = C.c_void_p()
        _check(self._lib.sr_device_create(hip_ordinal, C.byref(self._h)), "sr_device_create")
        self.ordinal = hip_ordinal

    def close(self):
        if self._h:
            self._lib.sr_device_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def synchronize(self):
        _check(self._lib.sr_device_synchronize(self._h), "sr_device_synchronize")

    wait_idle = synchronize  # vkh::GPUDevice::wait_idle, src/main.cpp:170

    def set_normal_mapping(self, enabled: bool):
        """Draws issued from now on shade with the normal map's normals (shaders/shader.frag:35-53 with the body the reference
        has commented out restored); off by default, as in the reference."""
        _check(self._lib.sr_device_set_normal_mapping(self._h, int(enabled)), "sr_device_set_normal_mapping")

    def draw_counts(self) -> dict:
        """How the draws of this device were issued so far: draws, lean_draws, lean_redraws, overflow_redraws (cumulative)."""
        c = DrawCounts()
        _check(self._lib.sr_device_get_draw_counts(self._h, C.byref(c)), "sr_device_get_draw_counts")
        return {n: int(getattr(c, n)) for n, _ in c._fields_}

    def profile(self, enabled: bool):
        _check(self._lib.sr_profile_enable(self._h, int(enabled)), "sr_profile_enable")

    def stage_times(self) -> dict:
        t = StageTimes()
        _check(self._lib.sr_profile_get(self._h, C.byref(t)), "sr_profile_get")
        return t.as_dict()


class StaticMesh:
    """Device-resident vertex and index buffers (src/mesh.hpp:63-67)."""

    def __init__(self, device: Device, handle, n_vertices: int, indices_size: int):
        self._dev, self._h = device, handle
        self.n_vertices, self.indices_size = n_vertices, indices_size

    def destroy(self):
        if self._h:
            self._dev._lib.sr_mesh_destroy(self._dev._h, self._h)
            self._h = None

    def vertex_stage(self, ubo52: np.ndarray, width: int, height: int):
        """(clip (n,4) f32, screen records (n,)) of the vertex stage for one uniform block."""
        ubo = np.ascontiguousarray(ubo52, np.float32).reshape(-1)
        assert ubo.size >= 51
        ubo = np.concatenate([ubo, np.zeros(max(0, 52 - ubo.size), np.float32)])
        clip = np.empty((self.n_vertices, 4), np.float32)
        screen = np.empty(self.n_vertices, SCREEN_DTYPE)
        _check(self._dev._lib.sr_debug_vertex_stage(self._dev._h, self._h, ubo.ctypes.data, width, height,
                                                     clip.ctypes.data, screen.ctypes.data), "sr_debug_vertex_stage")
        return clip, screen


def create_mesh_from_data(device: Device, vertices: np.ndarray, indices: np.ndarray) -> StaticMesh:
    """vertices: (n,8) float32 records of the reference's Vertex; indices: uint32 triangle list."""
    v = np.ascontiguousarray(vertices, np.float32)
    if v.ndim != 2 or v.shape[1] != 8:
        raise ValueError("vertices must have shape (n, 8): pos3, normal3, uv2")
    i = np.ascontiguousarray(indices, np.uint32).reshape(-1)
    h = C.c_void_p()
    _check(device._lib.sr_mesh_create(device._h, v.ctypes.data if v.size else None, v.shape[0],
                                      i.ctypes.data if i.size else None, i.size, C.byref(h)), "sr_mesh_create")
    return StaticMesh(device, h, v.shape[0], int(i.size))


class Texture:
    def __init__(self, device: Device, handle, width: int, height: int):
        self._dev, self._h = device, handle
        self.width, self.height = width, height
        self.mip_levels = device._lib.sr_texture_mip_levels(handle)

    def read_level(self, level: int) -> np.ndarray:
        w, h = max(1, self.width >> level), max(1, self.height >> level)
        out = np.empty((h, w, 4), np.uint8)
        _check(self._dev._lib.sr_texture_read_level(self._dev._h, self._h, level, out.ctypes.data, out.nbytes), "sr_texture_read_level")
        return out

    def destroy(self):
        if self._h:
            self._dev._lib.sr_texture_destroy(self._dev._h, self._h)
            self._h = None


def create_texture(device: Device, data: np.ndarray, tex_width: int | None = None, tex_height: int | None = None,
                   sampler: SamplerDesc | None = None) -> Texture:
    """data: RGBA8 pixels, (h,w,4) uint8 (src/texture.cpp:178-271)."""
    d = np.ascontiguousarray(data, np.uint8)
    if tex_width is None:
        tex_height, tex_width = d.shape[:2]
    if d.size != tex_width * tex_height * 4:
        raise ValueError("data must hold width*height RGBA8 texels")
    h = C.c_void_p()
    _check(device._lib.sr_texture_create_rgba8_srgb(device._h, d.ctypes.data, tex_width, tex_height, _sampler_arg(sampler), C.byref(h)),
           "sr_texture_create_rgba8_srgb")
    return Texture(device, h, tex_width, tex_height)


def destroy_texture(device: Device, texture: Texture):
    texture.destroy()


def read_image(path: str) -> np.ndarray:
    """Decodes an image file to (h,w,4) uint8 RGBA the way stbi_load(path, ..., STBI_rgb_alpha) expands it
    (src/texture.cpp:278-279).  Host only.  PNG and JPEG files."""
    L = load_library()
    p, w, h = C.c_void_p(), C.c_int(), C.c_int()
    _check(L.sr_image_read_file(os.fsencode(path), C.byref(p), C.byref(w), C.byref(h)), "sr_image_read_file")
    try:
        return np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_uint8)), shape=(h.value, w.value, 4)).copy()
    finally:
        L.sr_image_free(p)


def create_texture_from_file(device: Device, texture_path: str, sampler: SamplerDesc | None = None) -> Texture:
    """src/texture.hpp:20 / src/texture.cpp:273-291."""
    h = C.c_void_p()
    _check(device._lib.sr_texture_create_from_file(device._h, os.fsencode(texture_path), _sampler_arg(sampler), C.byref(h)), "sr_texture_create_from_file")
    w, ht = C.c_int(), C.c_int()
    _check(device._lib.sr_texture_size(h, C.byref(w), C.byref(ht)), "sr_texture_size")
    return Texture(device, h, w.value, ht.value)


def _model_arrays(L, handle):
    out = []
    for i in range(L.sr_model_mesh_count(handle)):
        v, nv, ix, ni = C.c_void_p(), C.c_uint32(), C.c_void_p(), C.c_uint32()
        _check(L.sr_model_mesh_data(handle, i, C.byref(v), C.byref(nv), C.byref(ix), C.byref(ni)), "sr_model_mesh_data")
        verts = np.ctypeslib.as_array(C.cast(v, C.POINTER(C.c_float)), shape=(nv.value, 8)).copy() if nv.value else np.zeros((0, 8), np.float32)
        idx = np.ctypeslib.as_array(C.cast(ix, C.POINTER(C.c_uint32)), shape=(ni.value,)).copy() if ni.value else np.zeros(0, np.uint32)
        out.append((verts, idx))
    return out


def read_model(path: str):
    """The reading half of Model.load, host only: a list of (vertices (n,8) f32, indices u32), one per
    primitive in node order."""
    L = load_library()
    h = C.c_void_p()
    _check(L.sr_model_read_file(os.fsencode(path), C.byref(h)), "sr_model_read_file")
    try:
        return _model_arrays(L, h)
    finally:
        L.sr_model_destroy(None, h)


class Model:
    """src/model.hpp:9-26: the meshes of a glTF file, uploaded."""

    def __init__(self, device: Device, handle):
        self._dev, self._h = device, handle
        L = device._lib
        self._meshes = []
        for i, (verts, idx) in enumerate(_model_arrays(L, handle)):
            m = StaticMesh(device, C.c_void_p(L.sr_model_mesh(handle, i)), verts.shape[0], int(idx.size))
            m.destroy = lambda: None   # owned by the model
            self._meshes.append(m)

    def meshes(self):
        return self._meshes

    @staticmethod
    def load(device: Device, path: str) -> "Model":
        h = C.c_void_p()
        _check(device._lib.sr_model_load(device._h, os.fsencode(path), C.byref(h)), "sr_model_load")
        return Model(device, h)

    def destroy(self):
        if self._h:
            self._dev._lib.sr_model_destroy(self._dev._h, self._h)
            self._h = None
            self._meshes = []


class RenderTarget:
    def __init__(self, device: Device, width: int, height: int, samples: int = 1, n_views: int = 1, rect=None):
        self._dev = device
        self.width, self.height, self.samples, self.n_views = width, height, samples, n_views
        self.rect = tuple(rect) if rect is not None else (0, 0, width, height)
        r = Rect(*self.rect)
        self._h = C.c_void_p()
        _check(device._lib.sr_target_create(device._h, width, height, samples, n_views, C.byref(r) if rect is not None else None,
                                            C.byref(self._h)), "sr_target_create")

    def destroy(self):
        if self._h:
            self._dev._lib.sr_target_destroy(self._dev._h, self._h)
            self._h = None

    def bind_color(self, device_ptr: int | None):
        _check(self._dev._lib.sr_target_bind_color(self._dev._h, self._h, C.c_void_p(device_ptr) if device_ptr else None), "sr_target_bind_color")

    def _read(self, kind, view, shape, dtype):
        out = np.empty(shape, dtype)
        _check(self._dev._lib.sr_target_readback(self._dev._h, self._h, kind, view, out.ctypes.data, out.nbytes), "sr_target_readback")
        return out

    def read_color(self, view: int = 0) -> np.ndarray:
        return self._read(READ_COLOR_BGRA8, view, (self.rect[3], self.rect[2], 4), np.uint8)

    def read_depth(self, view: int = 0) -> np.ndarray:
        return self._read(READ_DEPTH_F32, view, (self.rect[3], self.rect[2], self.samples), np.float32)

    def read_triangle_ids(self, view: int = 0) -> np.ndarray:
        return self._read(READ_TRI_ID_U32, view, (self.rect[3], self.rect[2], self.samples), np.uint32)

    def read_visibility(self, view: int = 0) -> np.ndarray:
        return self._read(READ_VIS_U64, view, (self.rect[3], self.rect[2], self.samples), np.uint64)

    def stats(self) -> dict:
        s = DrawStats()
        _check(self._dev._lib.sr_draw_get_stats(self._dev._h, self._h, C.byref(s)), "sr_draw_get_stats")
        return s.as_dict()


def draw(device: Device, target: RenderTarget, mesh: StaticMesh, ubo_blocks: np.ndarray, textures, no_alloc: bool = False):
    """ubo_blocks: (n_views, 52) float32 uniform blocks (or (52,) for one view); textures: 4 Texture
    objects in binding order albedo, normal, metallic, roughness (normal may be None).  Asynchronous.
    no_alloc: through sr_draw_no_alloc (every intermediate buffer in the library's static device array; at most 8 views)."""
    u = np.ascontiguousarray(ubo_blocks, np.float32)
    if u.ndim == 1:
        u = u.reshape(1, -1)
    if u.shape[1] < 51:
        raise ValueError("a uniform block has at least 51 floats (204 bytes)")
    if u.shape[1] != 52:
        u = np.ascontiguousarray(np.concatenate([u[:, :51], np.zeros((u.shape[0], 1), np.float32)], axis=1))
    arr = (C.c_void_p * 4)(*[t._h if t is not None else None for t in textures])
    entry = device._lib.sr_draw_no_alloc if no_alloc else device._lib.sr_draw
    _check(entry(device._h, target._h, mesh._h, u.ctypes.data, u.shape[0], u.shape[1] * 4, arr), "sr_draw_no_alloc" if no_alloc else "sr_draw")
