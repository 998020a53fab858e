"""Properties every rasteriser that follows the Vulkan fill rules must have, checked on the CPU oracle
with randomly drawn triangles (hypothesis).  They pin the oracle where no closed-form answer is at hand:
a mesh of triangles that share edges covers every sample exactly once (no cracks, no double hits), a
shift by whole pixels shifts the coverage, the frame does not depend on the triangle order, and a
restricted rectangle stores exactly the crop."""
import numpy as np
from hypothesis import given, settings, strategies as st

from softraster import scene
from test_oracle_known_answers import TEX, ortho_ubo, px_to_ndc

W, H = 48, 32
coord = st.floats(min_value=-6.0, max_value=54.0, allow_nan=False, width=32)


def fan_mesh(points, centre, z):
    """A fan of triangles around `centre` through `points` (window pixels), each oriented counter-clockwise
    as displayed; triangles that would be clockwise are flipped so that nothing is culled."""
    v = np.zeros((len(points) + 1, 8), np.float32)
    for i, (x, y) in enumerate([centre] + points):
        v[i, 0:2] = px_to_ndc(x, y, W, H)
        v[i, 2] = z
        v[i, 5] = 1.0
        v[i, 6:8] = (x / W, y / H)
    tris = []
    for i in range(len(points)):
        a, b, c = 0, 1 + i, 1 + (i + 1) % len(points)
        (ax, ay), (bx, by), (cx, cy) = centre, points[i], points[(i + 1) % len(points)]
        area = (bx - ax) * (cy - ay) - (cx - ax) * (by - ay)      # window coordinates, y down
        tris.append((a, b, c) if area < 0 else (a, c, b))          # counter-clockwise as displayed = negative area here
    return v, np.array(tris, np.uint32).reshape(-1)


@settings(max_examples=60, deadline=None)
@given(st.lists(st.tuples(coord, coord), min_size=3, max_size=7), st.tuples(coord, coord), st.sampled_from([1, 4]))
def test_shared_edges_are_covered_exactly_once(oracle, points, centre, samples):
    """Draw the triangles of a fan one by one and all together: every sample that the union covers is
    covered by exactly one triangle, unless two triangles of the fan genuinely overlap (a fan around a
    point outside the polygon folds over itself) -- so the check is per shared edge: two triangles on
    opposite sides of their common edge never both cover a sample."""
    ubo = ortho_ubo()
    v, idx = fan_mesh(points, centre, 0.5)
    tris = idx.reshape(-1, 3)
    cover = []
    for t in tris:
        r = oracle.draw(ubo, v, t.copy(), TEX, W, H, samples, shade=False)
        cover.append(r["tri"] != 0xFFFFFFFF)
    pos = v[:, 0:2].astype(np.float64)
    for i in range(len(tris)):
        for j in range(i + 1, len(tris)):
            shared = set(tris[i].tolist()) & set(tris[j].tolist())
            if len(shared) != 2:
                continue
            a, b = (pos[k] for k in shared)
            ci = pos[(set(tris[i].tolist()) - shared).pop()]
            cj = pos[(set(tris[j].tolist()) - shared).pop()]
            side = lambda p: (b[0] - a[0]) * (p[1] - a[1]) - (b[1] - a[1]) * (p[0] - a[0])
            if side(ci) * side(cj) < 0:     # opposite sides of the shared edge: interiors are disjoint
                assert not (cover[i] & cover[j]).any()


@settings(max_examples=40, deadline=None)
@given(st.lists(st.tuples(coord, coord), min_size=3, max_size=3), st.integers(-5, 5), st.integers(-5, 5), st.sampled_from([1, 4]))
def test_whole_pixel_shift_shifts_the_coverage(oracle, tri, dx, dy, samples):
    """Snapping is to 1/256 pixel, so moving a triangle whose vertices are multiples of 1/256 by whole pixels
    moves its coverage by whole pixels (inside the part of the frame both versions keep)."""
    ubo = ortho_ubo()
    quant = [(round(x * 256) / 256, round(y * 256) / 256) for x, y in tri]

    def draw(points):
        v = np.zeros((3, 8), np.float32)
        for i, (x, y) in enumerate(points):
            v[i, 0:2] = px_to_ndc(x, y, 64, 64)      # 64 x 64: pixel -> NDC is exact in float32 for multiples of 1/256
            v[i, 2] = 0.25
        area = (points[1][0] - points[0][0]) * (points[2][1] - points[0][1]) - (points[2][0] - points[0][0]) * (points[1][1] - points[0][1])
        idx = np.array([0, 1, 2] if area < 0 else [0, 2, 1], np.uint32)
        return oracle.draw(ubo, v, idx, TEX, 64, 64, samples, shade=False)["tri"] != 0xFFFFFFFF

    base = draw(quant)
    moved = draw([(x + dx, y + dy) for x, y in quant])
    ys, xs = slice(max(0, dy), 64 + min(0, dy)), slice(max(0, dx), 64 + min(0, dx))
    ys0, xs0 = slice(max(0, -dy), 64 + min(0, -dy)), slice(max(0, -dx), 64 + min(0, -dx))
    assert np.array_equal(moved[ys, xs], base[ys0, xs0])


@settings(max_examples=25, deadline=None)
@given(st.integers(0, 2 ** 31), st.sampled_from([1, 4]))
def test_order_and_rectangle_independence(oracle, seed, samples):
    import util
    verts, idx = util.random_soup(40, seed % 100000, size=0.9)
    ubo = scene.default_camera_ubo(W, H)
    full = oracle.draw(ubo, verts, idx, TEX, W, H, samples)
    perm = np.random.default_rng(seed).permutation(idx.size // 3)
    shuffled = oracle.draw(ubo, verts, np.ascontiguousarray(idx.reshape(-1, 3)[perm]).reshape(-1), TEX, W, H, samples)
    assert np.array_equal(full["depth"].view(np.uint32), shuffled["depth"].view(np.uint32))
    # the duplicates random_soup plants (every 7th triangle repeats its predecessor) tie on depth: the lower id wins
    tri_s = shuffled["tri"]
    mapped = np.where(tri_s == 0xFFFFFFFF, 0xFFFFFFFF, perm[np.minimum(tri_s, perm.size - 1)])
    differ = mapped != full["tri"]
    if differ.any():      # only ever between a triangle and its planted duplicate
        a, b = mapped[differ].astype(np.int64), full["tri"][differ].astype(np.int64)
        assert (np.abs(a - b) == 1).all() and ((np.maximum(a, b) % 7) == 0).all()
    rect = (5, 3, 31, 22)
    part = oracle.draw(ubo, verts, idx, TEX, W, H, samples, rect=rect)
    x, y, w, h = rect
    assert np.array_equal(part["tri"], full["tri"][y:y + h, x:x + w]) and np.array_equal(part["bgra"], full["bgra"][y:y + h, x:x + w])
