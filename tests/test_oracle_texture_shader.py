"""Known answers for the texture path (src/texture.cpp:87-176, 185-263) and the fragment shader
(shaders/shader.frag:16-152) of the CPU oracle."""
import math

import numpy as np
import pytest

from softraster import scene


def test_mip_level_count(oracle):
    # floor(log2(max(w, h))) + 1, src/texture.cpp:187-189: the reference's 2048^2 maps have 12 levels, a 4096^2 map 13
    assert oracle.mip_levels(2048, 2048) == 12 and oracle.mip_levels(4096, 4096) == 13
    assert oracle.mip_levels(1, 1) == 1 and oracle.mip_levels(5, 3) == 3 and oracle.mip_levels(100, 37) == 7
    assert oracle.mip_offset(4, 4, 1) == 16 and oracle.mip_offset(4, 4, 2) == 20 and oracle.mip_offset(4, 4, 3) == 21
    # total chain size of the reference's maps: 22.4 MB for 2048^2, 89.5 MB for 4096^2
    assert oracle.mip_offset(2048, 2048, 12) * 4 == 22369620 and oracle.mip_offset(4096, 4096, 13) * 4 == 89478484


def test_srgb_tables(oracle):
    dec, thr = oracle.srgb_tables()
    assert dec[0] == 0.0 and dec[255] == 1.0 and (np.diff(dec) > 0).all() and (np.diff(thr) > 0).all()
    assert dec[1] == pytest.approx(1 / 255 / 12.92, rel=1e-6)
    assert dec[128] == pytest.approx(((128 / 255 + 0.055) / 1.055) ** 2.4, rel=1e-6)
    for c in range(256):                       # encode is the exact inverse on the 256 codes
        assert oracle.srgb_encode8(dec[c]) == c
    for c in range(255):                       # and switches exactly at the half-way thresholds
        assert oracle.srgb_encode8(thr[c]) == c + 1
        assert oracle.srgb_encode8(np.nextafter(thr[c], np.float32(0))) == c
    assert oracle.srgb_encode8(-1.0) == 0 and oracle.srgb_encode8(float("nan")) == 0 and oracle.srgb_encode8(7.0) == 255


def test_mip_generation_is_a_box_filter_in_linear_light(oracle):
    img = np.zeros((2, 2, 4), np.uint8)
    img[0, 0] = (255, 0, 10, 255); img[0, 1] = (0, 0, 20, 255); img[1, 0] = (0, 255, 30, 0); img[1, 1] = (255, 255, 40, 0)
    chain = oracle.build_mip_chain(img)
    dec, _ = oracle.srgb_tables()
    assert chain.shape == (5, 4) and np.array_equal(chain[:4].reshape(2, 2, 4), img)
    top = chain[4]
    for c in range(3):
        mean = (dec[img[0, 0, c]] * 0.25 + dec[img[0, 1, c]] * 0.25 + dec[img[1, 0, c]] * 0.25) + dec[img[1, 1, c]] * 0.25
        assert top[c] == oracle.srgb_encode8(mean)
    assert top[0] == 188           # half of full-scale red in linear light is sRGB 188, not 128
    assert top[3] == 128           # alpha is linear: round-half-even of 127.5
    # odd sizes: a 5x1 strip goes to 2x1 by bilinear resampling at (i + 0.5) * 2.5
    strip = np.zeros((1, 5, 4), np.uint8); strip[0, :, 0] = (0, 50, 100, 150, 200); strip[..., 3] = 255
    ch = oracle.build_mip_chain(strip)
    l1 = oracle.chain_level(ch, 5, 1, 1)
    assert l1.shape == (1, 2, 4)
    assert l1[0, 0, 0] == oracle.srgb_encode8(0.25 * dec[0] + 0.75 * dec[50])      # u = 0.75
    assert l1[0, 1, 0] == oracle.srgb_encode8(0.75 * dec[150] + 0.25 * dec[200])   # u = 3.25


def test_bilinear_trilinear_and_repeat(oracle):
    img = np.zeros((2, 2, 4), np.uint8); img[..., 3] = 255
    img[0, 0, :3] = 255; img[1, 1, :3] = 255          # 2x2 checker
    chain = oracle.build_mip_chain(img)
    dec, _ = oracle.srgb_tables()
    # texel centres return the decoded texel, LOD 0 without derivatives
    for (s, t, want) in [(0.25, 0.25, 1.0), (0.75, 0.25, 0.0), (0.25, 0.75, 0.0), (0.75, 0.75, 1.0)]:
        rgba, lod = oracle.sample(chain, 2, 2, s, t)
        assert lod == 0.0 and rgba[0] == want and rgba[3] == 1.0
    assert oracle.sample(chain, 2, 2, 0.5, 0.5)[0][0] == 0.5          # centre of the footprint: mean of the four
    assert oracle.sample(chain, 2, 2, 0.5, 0.25)[0][0] == 0.5
    # REPEAT: the same value one period away, and across the seam (u = 0 blends texel 1 and texel 0)
    assert np.array_equal(oracle.sample(chain, 2, 2, 1.25, -0.75)[0], oracle.sample(chain, 2, 2, 0.25, 0.25)[0])
    assert oracle.sample(chain, 2, 2, 0.0, 0.25)[0][0] == 0.5
    # LOD from derivatives: one texel per pixel -> 0; two -> 1 (the 1x1 level); sqrt(2) -> 0.5
    top = dec[chain[4][0]]
    assert oracle.sample(chain, 2, 2, 0.25, 0.25, 0.5, 0, 0, 0.5)[1] == 0.0
    rgba, lod = oracle.sample(chain, 2, 2, 0.25, 0.25, 1.0, 0, 0, 1.0)
    assert lod == 1.0 and rgba[0] == top
    rgba, lod = oracle.sample(chain, 2, 2, 0.25, 0.25, math.sqrt(2) / 2, 0, 0, 0.1)
    assert lod == pytest.approx(0.5, abs=1e-6) and rgba[0] == pytest.approx(0.5 * 1.0 + 0.5 * top, abs=1e-6)
    # clamped to the last level however large the footprint; the longer axis decides
    assert oracle.sample(chain, 2, 2, 0.3, 0.3, 64.0, 0, 0, 0.001)[1] == 1.0


def shader_reference(wp, N, cam, albedo_tex, metallic, roughness):
    """shaders/shader.frag:102-152 in float64, written from the GLSL."""
    wp, N, cam = (np.asarray(v, np.float64) for v in (wp, N, cam))
    PI = 3.14159265359
    albedo = np.asarray(albedo_tex, np.float64) ** 2.2
    V = (cam - wp) / np.linalg.norm(cam - wp)
    F0 = 0.04 * (1 - metallic) + albedo * metallic
    Lo = np.zeros(3)
    for lp in [(-10, 10, 10), (10, 10, 10), (-10, -10, 10), (10, -10, 10)]:
        D = np.asarray(lp, np.float64) - wp
        L = D / np.linalg.norm(D)
        Hv = (V + L) / np.linalg.norm(V + L)
        radiance = np.array([23.47, 21.31, 20.79]) / D.dot(D)
        a2 = (roughness * roughness) ** 2
        NdotH = max(N.dot(Hv), 0.0)
        NDF = a2 / (PI * (NdotH * NdotH * (a2 - 1) + 1) ** 2)
        k = (roughness + 1) ** 2 / 8
        g = lambda c: c / (c * (1 - k) + k)
        NdotV, NdotL = max(N.dot(V), 0.0), max(N.dot(L), 0.0)
        G = g(NdotL) * g(NdotV)
        F = F0 + (1 - F0) * (1 - max(Hv.dot(V), 0.0)) ** 5
        spec = NDF * G * F / max(4 * NdotV * NdotL + 0.001, 0.001)
        kD = (1 - F) * (1 - metallic)
        Lo += (kD * albedo / PI + spec) * radiance * NdotL
    color = 0.03 * albedo * 0.01 + Lo
    color = color / (color + 1)
    return color ** (1 / 2.2)


@pytest.mark.parametrize("case", [
    dict(wp=(0.5, 0.5, 0.70710678), N=(0.5, 0.5, 0.70710678), albedo=(0.8, 0.3, 0.1), metallic=0.0, roughness=0.5),
    dict(wp=(0.0, 0.0, 1.0), N=(0.0, 0.0, 1.0), albedo=(1.0, 1.0, 1.0), metallic=1.0, roughness=0.3),
    dict(wp=(0.9, -0.3, 0.2), N=(0.45, -0.15, 0.1), albedo=(0.2, 0.6, 0.9), metallic=0.4, roughness=0.9),   # un-normalised N, as the shader uses it
    dict(wp=(-0.6, 0.0, -0.8), N=(-0.6, 0.0, -0.8), albedo=(0.5, 0.5, 0.5), metallic=0.2, roughness=0.28),  # facing away from the camera
])
def test_fragment_shader_closed_form(oracle, case):
    cam = (2.0, 2.0, 2.0)
    got = oracle.fragment_main(case["wp"], case["N"], cam, case["albedo"], case["metallic"], case["roughness"])
    want = shader_reference(case["wp"], case["N"], cam, case["albedo"], case["metallic"], case["roughness"])
    assert np.allclose(got, want, rtol=2e-5, atol=1e-6)


def test_constant_textures_give_the_shader_colour_everywhere(oracle):
    """A frame with constant maps: every covered pixel equals the shader evaluated with those constants
    at the interpolated position -- checks interpolation, sampling and the sRGB store together."""
    W, H = 64, 48
    v, i = scene.uv_sphere(48, 32)
    alb, met, rgh = (200, 120, 40, 255), 64, 180
    tex = [scene.constant_rgba(8, 8, alb), scene.constant_rgba(8, 8, (128, 128, 255, 255)),
           scene.constant_rgba(4, 4, (met,) * 3 + (255,)), scene.constant_rgba(16, 16, (rgh,) * 3 + (255,))]
    ubo = scene.default_camera_ubo(W, H)
    r = oracle.draw(ubo, v, i, tex, W, H)
    dec, _ = oracle.srgb_tables()
    y, x = H // 2, W // 2
    assert r["tri"][y, x, 0] != 0xFFFFFFFF
    # pixel centre -> ray -> unit sphere hit (the mesh is a fine sphere, so its surface point is within ~1e-3)
    view = ubo[16:32].reshape(4, 4).T.astype(np.float64); proj = ubo[32:48].reshape(4, 4).T.astype(np.float64)
    ndc = np.array([2 * (x + 0.5) / W - 1, 1 - 2 * (y + 0.5) / H])
    dir_cam = np.array([ndc[0] / proj[0, 0], ndc[1] / proj[1, 1], -1.0])
    d = view[:3, :3].T @ dir_cam; d /= np.linalg.norm(d)
    o = np.array([2.0, 2.0, 2.0])
    b = o.dot(d); t = -b - math.sqrt(b * b - (o.dot(o) - 1))
    hit = o + t * d
    want = shader_reference(hit, hit, o, [dec[c] for c in alb[:3]], dec[met], dec[rgh])
    got = r["bgra"][y, x, [2, 1, 0]].astype(int)
    for c in range(3):
        assert abs(int(got[c]) - oracle.srgb_encode8(min(want[c], 1.0))) <= 2   # facetted normal vs true sphere normal


def test_anisotropic_footprint_is_the_specification_s(oracle):
    """Switched on, the oracle's texture() follows the Vulkan specification's worked method: eta =
    min(rho_max / rho_min, 16), N = ceil(eta) trilinear samples at i/(N+1) - 1/2 along the major axis'
    derivative, level of detail log2(rho_max / eta).  Checked by rebuilding the result from isotropic
    single-level samples (zero derivatives select level 0; a tap at level k is taken from a texture whose
    level 0 is that level)."""
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, size=(64, 64, 4), dtype=np.uint8)
    img[..., 3] = 255
    chain = oracle.build_mip_chain(img)
    levels = [oracle.chain_level(chain, 64, 64, l) for l in range(7)]
    level_chains = [oracle.build_mip_chain(np.ascontiguousarray(l)) for l in levels]

    def at_level(k, s, t):   # bilinear sample of level k alone
        side = 64 >> k
        return oracle.sample(level_chains[k], side, side, s, t)[0]

    s0, t0 = 0.37, 0.61
    try:
        oracle.set_anisotropic(True)
        for (dsdx, dtdx, dsdy, dtdy) in [(8 / 64, 0, 0, 0.5 / 64),          # ratio 16: N = 16, lambda = log2(8/16) -> level 0
                                         (0, 0.4 / 64, 3.2 * 0.4 / 64 * 4, 0),  # y major, ratio 12.8: N = 13, lambda = log2(5.12/12.8) -> level 0
                                         (12 / 64, 0, 0, 3 / 64),            # ratio 4: N = 4, lambda = log2(3) = 1.585 -> levels 1 and 2
                                         (40 / 64, 0, 0, 1 / 64)]:           # ratio 40 capped to 16: lambda = log2(2.5) = 1.32
            got, _ = oracle.sample(chain, 64, 64, s0, t0, dsdx, dtdx, dsdy, dtdy)
            rx, ry = np.hypot(dsdx * 64, dtdx * 64), np.hypot(dsdy * 64, dtdy * 64)
            rmax, rmin = max(rx, ry), min(rx, ry)
            eta = min(rmax / rmin, 16.0)
            n = int(np.ceil(np.float32(eta)))
            lam = max(float(np.log2(np.float32(rmax) / np.float32(eta))), 0.0)
            lo, frac = int(np.floor(lam)), np.float32(lam) - np.floor(np.float32(lam))
            ds, dt = (dsdx, dtdx) if rx > ry else (dsdy, dtdy)
            acc = np.zeros(4, np.float32)
            for i in range(1, n + 1):
                off = np.float32(i) / np.float32(n + 1) - np.float32(0.5)
                s, t = np.float32(s0) + off * np.float32(ds), np.float32(t0) + off * np.float32(dt)
                tap = at_level(lo, s, t)
                if frac != 0:
                    tap = (np.float32(1) - frac) * tap + frac * at_level(lo + 1, s, t)
                acc += tap
            assert np.allclose(got, acc / np.float32(n), atol=2e-6), (dsdx, dsdy, got, acc / n)
        # equal axes: one sample, identical to the isotropic result
        iso_like, _ = oracle.sample(chain, 64, 64, s0, t0, 2 / 64, 0, 0, 2 / 64)
        oracle.set_anisotropic(False)
        iso, _ = oracle.sample(chain, 64, 64, s0, t0, 2 / 64, 0, 0, 2 / 64)
        assert np.array_equal(iso_like, iso)
    finally:
        oracle.set_anisotropic(False)


def test_normal_map_tangent_frame(oracle):
    """Normal mapping (shaders/shader.frag:35-53 with the commented-out body restored; off by default): a quad in the plane
    z = 0 under a constant normal map.  The expected normal is rebuilt in float64 from ray casts through the pixel centres of
    the 2x2 quad (dFdx / dFdy of WorldPos and TexCoord), the tangent frame T, B, N of the shader and the decoded texel; the
    frame's colour must be the shader's for that normal.  With the switch off the same frame is shaded with the vertex normal."""
    W, H = 64, 48
    half = 1.5
    v, i = scene.screen_quad(z=0.0, half=half, uv_scale=1.0)
    alb, met, rgh, nrm = (200, 120, 40, 255), 64, 180, (230, 90, 200, 255)
    tex = [scene.constant_rgba(8, 8, alb), scene.constant_rgba(8, 8, nrm),
           scene.constant_rgba(4, 4, (met,) * 3 + (255,)), scene.constant_rgba(16, 16, (rgh,) * 3 + (255,))]
    ubo = scene.default_camera_ubo(W, H)
    plain = oracle.draw(ubo, v, i, tex, W, H)
    oracle.set_normal_mapping(True)
    try:
        mapped = oracle.draw(ubo, v, i, tex, W, H)
    finally:
        oracle.set_normal_mapping(False)
    assert np.array_equal(plain["tri"], mapped["tri"]) and np.array_equal(plain["depth"], mapped["depth"])
    dec, _ = oracle.srgb_tables()
    view = ubo[16:32].reshape(4, 4).T.astype(np.float64); proj = ubo[32:48].reshape(4, 4).T.astype(np.float64)
    eye = np.array([2.0, 2.0, 2.0])

    def hit(px, py):   # pixel centre -> ray -> plane z = 0
        ndc = np.array([2 * (px + 0.5) / W - 1, 1 - 2 * (py + 0.5) / H])
        d = view[:3, :3].T @ np.array([ndc[0] / proj[0, 0], ndc[1] / proj[1, 1], -1.0])
        return eye + (-eye[2] / d[2]) * d

    uv = lambda p: (p[:2] + half) / (2 * half)
    tn = np.array([dec[c] for c in nrm[:3]], np.float64) * 2 - 1
    checked = 0
    for (x, y) in [(W // 2, H // 2), (W // 2 - 5, H // 2 + 4), (W // 2 + 7, H // 2 - 3), (W // 2 + 2, H // 2 + 7)]:
        if mapped["tri"][y, x, 0] == 0xFFFFFFFF:
            continue
        qx, qy = x & ~1, y & ~1
        p00, p10, p01 = hit(qx, qy), hit(qx + 1, qy), hit(qx, qy + 1)
        Q1, Q2, st1, st2 = p10 - p00, p01 - p00, uv(p10) - uv(p00), uv(p01) - uv(p00)
        N = np.array([0.0, 0.0, 1.0])
        T = Q1 * st2[1] - Q2 * st1[1]; T /= np.linalg.norm(T)
        B = np.cross(N, T); B = -B / np.linalg.norm(B)
        M = T * tn[0] + B * tn[1] + N * tn[2]; M /= np.linalg.norm(M)
        for frame, normal in ((mapped, M), (plain, N)):
            want = shader_reference(hit(x, y), normal, eye, [dec[c] for c in alb[:3]], dec[met], dec[rgh])
            got = frame["bgra"][y, x, [2, 1, 0]].astype(int)
            for c in range(3):
                assert abs(int(got[c]) - oracle.srgb_encode8(min(want[c], 1.0))) <= 1, (x, y, c, got, want)
        checked += 1
    assert checked >= 3
    assert (np.abs(mapped["bgra"].astype(int) - plain["bgra"].astype(int)).max(axis=-1) > 8).mean() > 0.2   # the map does change the frame
