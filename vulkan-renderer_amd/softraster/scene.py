"""Synthetic scenes for the draw path: meshes, textures, cameras and the uniform block.

Everything here is host-side input preparation (numpy only).  The vertex record is the
reference's ``Vertex`` (src/mesh.hpp:18-54): 8 float32 = pos3, normal3, uv2.  The uniform block is
the reference's ``UniformBufferObject`` (src/main.cpp:50-55) in its std140 form
(shaders/shader.vert:4-9): three column-major mat4 followed by a vec3, padded to 208 bytes.
"""
from __future__ import annotations

import math

import numpy as np

UBO_BYTES = 208
UBO_FLOATS = 52


# ----------------------------------------------------------------------------- cameras
def _normalize(v):
    v = np.asarray(v, dtype=np.float32)
    return (v / np.float32(np.sqrt(np.float32(np.dot(v, v))))).astype(np.float32)


def look_at(eye, center, up) -> np.ndarray:
    """GLM ``lookAt`` (right-handed) as called at src/main.cpp:992.  Returns a 4x4 array indexed
    [row, col]; use :func:`make_ubo` to serialise it column-major."""
    eye = np.asarray(eye, np.float32)
    center = np.asarray(center, np.float32)
    up = np.asarray(up, np.float32)
    f = _normalize(center - eye)
    s = _normalize(np.cross(f, up).astype(np.float32))
    u = np.cross(s, f).astype(np.float32)
    m = np.eye(4, dtype=np.float32)
    m[0, :3] = s
    m[1, :3] = u
    m[2, :3] = -f
    m[0, 3] = -np.dot(s, eye)
    m[1, 3] = -np.dot(u, eye)
    m[2, 3] = np.dot(f, eye)
    return m


def perspective(fovy_rad: float, aspect: float, z_near: float, z_far: float) -> np.ndarray:
    """GLM ``perspective`` with GLM_FORCE_DEPTH_ZERO_TO_ONE, right-handed (src/main.cpp:4-5, 993-997)."""
    f32 = np.float32
    t = f32(math.tan(f32(fovy_rad) / f32(2.0)))
    m = np.zeros((4, 4), dtype=np.float32)
    m[0, 0] = f32(1.0) / (f32(aspect) * t)
    m[1, 1] = f32(1.0) / t
    m[2, 2] = f32(z_far) / (f32(z_near) - f32(z_far))
    m[3, 2] = f32(-1.0)
    m[2, 3] = -(f32(z_far) * f32(z_near)) / (f32(z_far) - f32(z_near))
    return m


def make_ubo(model: np.ndarray, view: np.ndarray, proj: np.ndarray, camera_pos) -> np.ndarray:
    """Serialise one uniform block: 52 float32 (208 bytes), matrices column-major."""
    out = np.zeros(UBO_FLOATS, dtype=np.float32)
    out[0:16] = np.asarray(model, np.float32).T.reshape(-1)
    out[16:32] = np.asarray(view, np.float32).T.reshape(-1)
    out[32:48] = np.asarray(proj, np.float32).T.reshape(-1)
    out[48:51] = np.asarray(camera_pos, np.float32)
    return out


def default_camera_ubo(width: int, height: int, eye=(2.0, 2.0, 2.0), up=(0.0, 0.0, 1.0),
                       model: np.ndarray | None = None) -> np.ndarray:
    """The reference's per-frame block: identity model, eye (2,2,2) looking at the origin with +Z up,
    45 degree vertical field of view, near 0.1, far 10 (src/main.cpp:210-211, 988-1004)."""
    m = np.eye(4, dtype=np.float32) if model is None else model
    v = look_at(eye, (0.0, 0.0, 0.0), up)
    p = perspective(math.radians(45.0), float(width) / float(height), 0.1, 10.0)
    return make_ubo(m, v, p, eye)


def rotate_vector(v, angle: float, normal) -> np.ndarray:
    """glm::rotate(vec3, angle, normal) of glm/gtx/rotate_vector.hpp in float32: mat3(glm::rotate(angle, normal)) * v."""
    f = np.float32
    v = np.asarray(v, f)
    a = _normalize(np.asarray(normal, f)).astype(f)
    c, s = f(math.cos(float(f(angle)))), f(math.sin(float(f(angle))))
    t = (f(1.0) - c) * a
    col0 = np.array([c + t[0] * a[0], t[0] * a[1] + s * a[2], t[0] * a[2] - s * a[1]], f)
    col1 = np.array([t[1] * a[0] - s * a[2], c + t[1] * a[1], t[1] * a[2] + s * a[0]], f)
    col2 = np.array([t[2] * a[0] + s * a[1], t[2] * a[1] - s * a[0], c + t[2] * a[2]], f)
    return ((col0 * v[0] + col1 * v[1]) + col2 * v[2]).astype(f)


class OrbitCamera:
    """The camera of the reference's Application: eye (2,2,2) and up +Z (src/main.cpp:210-211), moved by the
    W/S/A/D keys in steps of 0.1 rad (key_callback, src/main.cpp:223-267)."""

    def __init__(self):
        self.camera_pos = np.array([2.0, 2.0, 2.0], np.float32)
        self.up = np.array([0.0, 0.0, 1.0], np.float32)

    def key(self, k: str) -> None:
        """One press (or repeat) of a key; keys other than W, S, A, D do nothing."""
        step = 0.1
        if k in ("W", "S"):    # over the top: eye and up turn about the axis perpendicular to both
            axis = _normalize(np.cross(self.camera_pos, self.up).astype(np.float32))
            angle = step if k == "W" else -step
            self.camera_pos = rotate_vector(self.camera_pos, angle, axis)
            self.up = rotate_vector(self.up, angle, axis)
        elif k in ("A", "D"):
            self.camera_pos = rotate_vector(self.camera_pos, step if k == "A" else -step, self.up)

    def keys(self, sequence: str) -> "OrbitCamera":
        for k in sequence:
            self.key(k)
        return self

    def ubo(self, width: int, height: int) -> np.ndarray:
        """update_uniform_buffer, src/main.cpp:988-1004"""
        return default_camera_ubo(width, height, eye=self.camera_pos, up=self.up)


def _rot_axis(axis, angle):
    a = _normalize(axis).astype(np.float64)
    c, s = math.cos(angle), math.sin(angle)
    x, y, z = a
    return np.array([[c + x * x * (1 - c), x * y * (1 - c) - z * s, x * z * (1 - c) + y * s],
                     [y * x * (1 - c) + z * s, c + y * y * (1 - c), y * z * (1 - c) - x * s],
                     [z * x * (1 - c) - y * s, z * y * (1 - c) + x * s, c + z * z * (1 - c)]])


def orbit_eye(k: int, n: int) -> np.ndarray:
    """View k of n on the orbit the reference's key handler generates (src/main.cpp:238-258): the
    default eye tilted about normalize(cross(eye, up)) by 0.35*sin(2*pi*k/7), then turned about +Z."""
    eye0 = np.array([2.0, 2.0, 2.0])
    axis = np.cross(eye0, [0.0, 0.0, 1.0])
    tilt = _rot_axis(axis, 0.35 * math.sin(2.0 * math.pi * k / 7.0))
    turn = _rot_axis([0.0, 0.0, 1.0], 2.0 * math.pi * k / n)
    return (turn @ (tilt @ eye0)).astype(np.float32)


def orbit_ubos(n_views: int, width: int, height: int, first: int = 0, total: int | None = None) -> np.ndarray:
    """(n_views, 52) float32 uniform blocks for views first..first+n_views-1 of a `total`-view orbit."""
    total = n_views if total is None else total
    return np.stack([default_camera_ubo(width, height, eye=orbit_eye(first + k, total)) for k in range(n_views)])


# ----------------------------------------------------------------------------- meshes
def uv_sphere(x_segments: int, y_segments: int):
    """z-up unit UV sphere with the parameterisation of src/main.cpp:94-115, indexed as a triangle
    list whose triangles are counter-clockwise seen from outside.  Returns (verts (V,8) f32,
    indices (3T,) u32) with V=(X+1)(Y+1), T=2XY.  Pole triangles are degenerate on purpose."""
    X, Y = int(x_segments), int(y_segments)
    ys = np.arange(Y + 1, dtype=np.float64) / Y
    xs = np.arange(X + 1, dtype=np.float64) / X
    theta = (ys * math.pi)[:, None]
    phi = (xs * 2.0 * math.pi)[None, :]
    px = np.cos(phi) * np.sin(theta)
    py = np.sin(phi) * np.sin(theta)
    pz = np.cos(theta) * np.ones_like(phi)
    verts = np.empty(((Y + 1) * (X + 1), 8), dtype=np.float32)
    verts[:, 0] = px.reshape(-1)
    verts[:, 1] = py.reshape(-1)
    verts[:, 2] = pz.reshape(-1)
    verts[:, 3:6] = verts[:, 0:3]
    verts[:, 6] = np.broadcast_to(xs[None, :], (Y + 1, X + 1)).reshape(-1)
    verts[:, 7] = np.broadcast_to(ys[:, None], (Y + 1, X + 1)).reshape(-1)
    yy, xx = np.meshgrid(np.arange(Y, dtype=np.int64), np.arange(X, dtype=np.int64), indexing="ij")
    i00 = yy * (X + 1) + xx
    i10 = i00 + 1            # next x
    i01 = i00 + (X + 1)      # next y
    i11 = i01 + 1
    tris = np.stack([i00, i01, i11, i00, i11, i10], axis=-1).reshape(-1)
    return verts, tris.astype(np.uint32)


def screen_quad(z: float = 0.0, half: float = 1.0, uv_scale: float = 1.0):
    """Two triangles in the z=const plane spanning [-half,half]^2, facing +Z."""
    h = half
    verts = np.array([[-h, -h, z, 0, 0, 1, 0.0, 0.0],
                      [h, -h, z, 0, 0, 1, uv_scale, 0.0],
                      [h, h, z, 0, 0, 1, uv_scale, uv_scale],
                      [-h, h, z, 0, 0, 1, 0.0, uv_scale]], dtype=np.float32)
    idx = np.array([0, 1, 2, 0, 2, 3], dtype=np.uint32)
    return verts, idx


# ----------------------------------------------------------------------------- textures
def albedo_pattern(size: int) -> np.ndarray:
    """(size,size,4) u8 procedural colour pattern: R=(x^y)&255, G=(3x+5y)&255, B=((x>>4)^(y>>4))*17&255."""
    y, x = np.meshgrid(np.arange(size, dtype=np.int64), np.arange(size, dtype=np.int64), indexing="ij")
    out = np.empty((size, size, 4), dtype=np.uint8)
    out[..., 0] = (x ^ y) & 255
    out[..., 1] = (x * 3 + y * 5) & 255
    out[..., 2] = (((x >> 4) ^ (y >> 4)) * 17) & 255
    out[..., 3] = 255
    return out


def noise_rgba(width: int, height: int, seed: int = 0) -> np.ndarray:
    rng = np.random.default_rng(seed)
    out = rng.integers(0, 256, size=(height, width, 4), dtype=np.uint8)
    out[..., 3] = 255
    return out


def gray_map(width: int, height: int, lo: int, hi: int, seed: int | None = None) -> np.ndarray:
    """Gray RGBA8 map (R=G=B, A=255) the way stb expands a one-channel PNG under STBI_rgb_alpha
    (src/texture.cpp:278-279): a diagonal ramp over [lo,hi], optionally with seeded noise."""
    y, x = np.meshgrid(np.arange(height, dtype=np.int64), np.arange(width, dtype=np.int64), indexing="ij")
    g = lo + ((x + 2 * y) % (hi - lo + 1))
    if seed is not None:
        rng = np.random.default_rng(seed)
        g = np.clip(g + rng.integers(-8, 9, size=g.shape), lo, hi)
    out = np.empty((height, width, 4), dtype=np.uint8)
    out[..., 0] = out[..., 1] = out[..., 2] = g.astype(np.uint8)
    out[..., 3] = 255
    return out


def constant_rgba(width: int, height: int, rgba) -> np.ndarray:
    out = np.empty((height, width, 4), dtype=np.uint8)
    out[...] = np.asarray(rgba, dtype=np.uint8)
    return out


def material_textures(size: int, seed: int | None = None):
    """[albedo, normal, metallic, roughness] in the binding order of src/main.cpp:461-516.  Metallic
    and roughness use the value ranges of the reference's bundled maps (0-254, 72-255)."""
    albedo = albedo_pattern(size) if seed is None else noise_rgba(size, size, seed)
    normal = constant_rgba(size, size, (128, 128, 255, 255))
    metallic = gray_map(size, size, 0, 254, seed)
    roughness = gray_map(size, size, 72, 255, None if seed is None else seed + 1)
    return [albedo, normal, metallic, roughness]
