"""ctypes binding of the CPU restatement (oracle/sr_oracle.c).

TEST INFRASTRUCTURE: imported only by tests/, the smoke check and the benchmark's cpu_baseline
leg.  The shipped library and its Python host module never import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "sr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"])
    return _LIB_PATH


class _DrawArgs(C.Structure):
    _fields_ = [
        ("ubo52", C.c_void_p),
        ("verts", C.c_void_p), ("n_verts", C.c_uint32),
        ("indices", C.c_void_p), ("n_indices", C.c_uint32),
        ("tex_chain", C.c_void_p * 4), ("tex_w", C.c_int * 4), ("tex_h", C.c_int * 4),
        ("W", C.c_int), ("H", C.c_int), ("S", C.c_int),
        ("rx", C.c_int), ("ry", C.c_int), ("rw", C.c_int), ("rh", C.c_int),
        ("threads", C.c_int),
        ("depth_out", C.c_void_p), ("tri_out", C.c_void_p), ("bgra_out", C.c_void_p), ("stats_out", C.c_void_p),
    ]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.sro_mip_levels.restype = C.c_int
        L.sro_mip_levels.argtypes = [C.c_int, C.c_int]
        L.sro_mip_offset.restype = C.c_size_t
        L.sro_mip_offset.argtypes = [C.c_int, C.c_int, C.c_int]
        L.sro_draw.restype = C.c_int
        L.sro_draw.argtypes = [C.POINTER(_DrawArgs)]
        L.sro_draw_args_size.restype = C.c_size_t
        L.sro_sample.argtypes = [C.c_void_p, C.c_int, C.c_int] + [C.c_float] * 6 + [C.c_void_p, C.c_void_p]
        L.sro_fragment_main.argtypes = [C.c_void_p] * 4 + [C.c_float, C.c_float, C.c_void_p]
        L.sro_perspective.argtypes = [C.c_float] * 4 + [C.c_void_p]
        L.sro_srgb_encode8.argtypes = [C.c_float]
        L.sro_srgb_encode8.restype = C.c_int
        assert L.sro_draw_args_size() == C.sizeof(_DrawArgs)
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_near_clip(on: bool):
    """Switches the near-plane clipper on or off for every later draw (off by default: triangles with a vertex that
    cannot be projected are dropped and counted)."""
    lib().sro_set_near_clip(1 if on else 0)


def set_anisotropic(on: bool):
    """Switches the specification's anisotropic footprint on or off for every later sample (off by default)."""
    lib().sro_set_anisotropic(1 if on else 0)


def set_normal_mapping(on: bool):
    """Switches the normal map on or off for every later draw (off by default: shaders/shader.frag:35-53 returns the interpolated
    normal, the body that reads the map is commented out in the reference)."""
    lib().sro_set_normal_mapping(1 if on else 0)


def srgb_tables():
    dec = np.empty(256, np.float32)
    thr = np.empty(255, np.float32)
    lib().sro_srgb_tables(_p(dec), _p(thr))
    return dec, thr


def srgb_encode8(x: float) -> int:
    return lib().sro_srgb_encode8(float(x))


def look_at(eye, center, up) -> np.ndarray:
    """column-major float32[16]"""
    m = np.empty(16, np.float32)
    e, c, u = (np.ascontiguousarray(v, np.float32) for v in (eye, center, up))
    lib().sro_look_at(_p(e), _p(c), _p(u), _p(m))
    return m


def perspective(fovy, aspect, zn, zf) -> np.ndarray:
    m = np.empty(16, np.float32)
    lib().sro_perspective(fovy, aspect, zn, zf, _p(m))
    return m


def mip_levels(w, h) -> int:
    return lib().sro_mip_levels(w, h)


def mip_offset(w, h, level) -> int:
    return lib().sro_mip_offset(w, h, level)


def build_mip_chain(rgba8: np.ndarray) -> np.ndarray:
    """rgba8: (h,w,4) u8.  Returns the flat chain (total_texels,4) u8, level 0 first."""
    h, w = rgba8.shape[:2]
    src = np.ascontiguousarray(rgba8, np.uint8)
    total = mip_offset(w, h, mip_levels(w, h))
    chain = np.empty((total, 4), np.uint8)
    lib().sro_build_mip_chain(_p(src), w, h, _p(chain))
    return chain


def chain_level(chain: np.ndarray, w: int, h: int, level: int) -> np.ndarray:
    o = mip_offset(w, h, level)
    lw, lh = max(1, w >> level), max(1, h >> level)
    return chain[o:o + lw * lh].reshape(lh, lw, 4)


def sample(chain: np.ndarray, w: int, h: int, s, t, dsdx=0.0, dtdx=0.0, dsdy=0.0, dtdy=0.0):
    out = np.empty(4, np.float32)
    lod = np.empty(1, np.float32)
    lib().sro_sample(_p(chain), w, h, s, t, dsdx, dtdx, dsdy, dtdy, _p(out), _p(lod))
    return out, float(lod[0])


def fragment_main(world_pos, normal, cam, albedo_tex_rgb, metallic, roughness) -> np.ndarray:
    a = [np.ascontiguousarray(v, np.float32) for v in (world_pos, normal, cam, albedo_tex_rgb)]
    out = np.empty(3, np.float32)
    lib().sro_fragment_main(_p(a[0]), _p(a[1]), _p(a[2]), _p(a[3]), metallic, roughness, _p(out))
    return out


def vertex_stage(ubo52: np.ndarray, verts: np.ndarray, width: int, height: int):
    """Returns (clip (n,4) f32, screen (n,) structured {X i4, Y i4, z f4, rw f4})."""
    verts = np.ascontiguousarray(verts, np.float32)
    ubo = np.ascontiguousarray(ubo52, np.float32)
    n = verts.shape[0]
    clip = np.empty((n, 4), np.float32)
    screen = np.empty(n, dtype=SCREEN_DTYPE)
    lib().sro_vertex_stage(_p(ubo), _p(verts), n, width, height, _p(clip), _p(screen))
    return clip, screen


SCREEN_DTYPE = np.dtype([("X", "<i4"), ("Y", "<i4"), ("z", "<f4"), ("rw", "<f4")])

STAT_NAMES = ("tris_in", "tris_front", "tris_clip_culled", "samples_tested", "samples_passed", "shaded_px", "tris_clipped")


def draw(ubo52, verts, indices, textures, width, height, samples=1, rect=None, threads=1, shade=True):
    """Render one view.  textures: list of 4 (h,w,4) u8 images (albedo, normal, metallic, roughness)
    or prebuilt (chain, w, h) tuples.  Returns dict(depth, tri, bgra, stats)."""
    ubo = np.ascontiguousarray(ubo52, np.float32).reshape(-1)[:52].copy()
    verts = np.ascontiguousarray(verts, np.float32)
    indices = np.ascontiguousarray(indices, np.uint32)
    rx, ry, rw, rh = rect if rect is not None else (0, 0, width, height)
    a = _DrawArgs()
    keep = []
    a.ubo52 = ubo.ctypes.data
    a.verts = verts.ctypes.data
    a.n_verts = verts.shape[0]
    a.indices = indices.ctypes.data
    a.n_indices = indices.size
    for k in range(4):
        t = textures[k]
        if isinstance(t, tuple):
            chain, tw, th = t
        else:
            th, tw = t.shape[:2]
            chain = build_mip_chain(t)
        keep.append(chain)
        a.tex_chain[k] = chain.ctypes.data
        a.tex_w[k] = tw
        a.tex_h[k] = th
    a.W, a.H, a.S = width, height, samples
    a.rx, a.ry, a.rw, a.rh = rx, ry, rw, rh
    a.threads = threads
    depth = np.empty((rh, rw, samples), np.float32)
    tri = np.empty((rh, rw, samples), np.uint32)
    bgra = np.empty((rh, rw, 4), np.uint8) if shade else None
    stats = np.zeros(8, np.uint64)
    a.depth_out = depth.ctypes.data
    a.tri_out = tri.ctypes.data
    a.bgra_out = bgra.ctypes.data if shade else None
    a.stats_out = stats.ctypes.data
    rc = lib().sro_draw(C.byref(a))
    if rc != 0:
        raise ValueError(f"sro_draw failed with {rc}")
    return {"depth": depth, "tri": tri, "bgra": bgra, "stats": dict(zip(STAT_NAMES, (int(v) for v in stats[:7])))}
