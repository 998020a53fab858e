"""File readers behind create_texture_from_file and Model::load (host-only entry points of the C ABI;
no GPU needed).  PNG: files written here in every colour type / bit depth / filter, decoded by the
library and by Pillow (what stb_image's STBI_rgb_alpha expansion amounts to); and the two PNGs the
reference ships, read where they lie.  glTF: files written here (data: uri, external .bin, .glb; triangle
list, strip and fan; strided and normalised accessors) and read back as the arrays that went in."""
import base64
import json
import os
import struct
import zlib

import numpy as np
import pytest

REF_TEXTURES = "/root/reference/data/textures/rustediron1-alt2-bl"


@pytest.fixture(scope="module")
def sr():
    import softraster
    softraster.load_library()
    return softraster


# ----------------------------------------------------------------------------------------------- PNG
def _chunk(tag, body):
    return struct.pack(">I", len(body)) + tag + body + struct.pack(">I", zlib.crc32(tag + body) & 0xFFFFFFFF)


def _paeth(a, b, c):
    p = a + b - c
    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
    return a if pa <= pb and pa <= pc else (b if pb <= pc else c)


def write_png(path, rows, width, height, ctype, depth, filters, plte=None, trns=None):
    """rows: list of bytes objects (already packed scan lines); filters: one filter type per row."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(channels * depth // 8, 1)
    raw = bytearray()
    prev = bytes(len(rows[0]))
    for y, row in enumerate(rows):
        f = filters[y % len(filters)]
        out = bytearray(len(row))
        for i, v in enumerate(row):
            a = row[i - bpp] if i >= bpp else 0
            b = prev[i]
            c = prev[i - bpp] if i >= bpp else 0
            pred = [0, a, b, (a + b) >> 1, _paeth(a, b, c)][f]
            out[i] = (v - pred) & 255
        raw += bytes([f]) + out
        prev = row
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0))
    if plte is not None:
        data += _chunk(b"PLTE", bytes(plte))
    if trns is not None:
        data += _chunk(b"tRNS", bytes(trns))
    z = zlib.compress(bytes(raw), 6)
    half = len(z) // 2
    data += _chunk(b"IDAT", z[:half]) + _chunk(b"IDAT", z[half:]) + _chunk(b"tEXt", b"Comment\x00two IDAT chunks") + _chunk(b"IEND", b"")
    with open(path, "wb") as f:
        f.write(data)


def _pack_rows(samples, depth):
    """samples: (h, n) integer array of raw sample values -> list of packed scan lines."""
    rows = []
    for r in samples:
        if depth == 8:
            rows.append(bytes(r.astype(np.uint8)))
        elif depth == 16:
            rows.append(r.astype(">u2").tobytes())
        else:
            bits = np.unpackbits(r.astype(np.uint8)[:, None], axis=1)[:, 8 - depth:].reshape(-1)
            rows.append(np.packbits(bits).tobytes())
    return rows


PNG_CASES = [(0, 1), (0, 2), (0, 4), (0, 8), (0, 16), (2, 8), (2, 16), (3, 1), (3, 2), (3, 4), (3, 8), (4, 8), (4, 16), (6, 8), (6, 16)]


@pytest.mark.parametrize("ctype,depth", PNG_CASES)
def test_png_matches_pillow(sr, tmp_path, ctype, depth):
    from PIL import Image
    rng = np.random.default_rng(ctype * 100 + depth)
    w, h = 37, 23   # odd sizes: partial bytes at low bit depths
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    samples = rng.integers(0, 1 << depth, size=(h, w * channels))
    samples[5:9] = samples[4]   # repeated rows make the Up and Paeth filters earn their keep
    plte = trns = None
    if ctype == 3:
        plte = rng.integers(0, 256, size=3 * (1 << depth)).tolist()
        trns = rng.integers(0, 256, size=(1 << depth) // 2 + 1).tolist()
    path = str(tmp_path / f"t_{ctype}_{depth}.png")
    write_png(path, _pack_rows(samples, depth), w, h, ctype, depth, filters=[0, 1, 2, 3, 4, 4, 3], plte=plte, trns=trns)
    got = sr.read_image(path)
    assert got.shape == (h, w, 4)
    if depth == 16:
        # stb keeps the high byte; Pillow has no uniform 16-bit RGBA path, so state the expectation directly
        hi = (samples >> 8).astype(np.uint8).reshape(h, w, channels)
        exp = np.full((h, w, 4), 255, np.uint8)
        if ctype == 0:
            exp[..., 0] = exp[..., 1] = exp[..., 2] = hi[..., 0]
        elif ctype == 4:
            exp[..., 0] = exp[..., 1] = exp[..., 2] = hi[..., 0]; exp[..., 3] = hi[..., 1]
        else:
            exp[..., :channels] = hi
    else:
        exp = np.asarray(Image.open(path).convert("RGBA"))
    assert np.array_equal(got, exp)


def test_png_colour_key_and_errors(sr, tmp_path):
    w, h = 8, 4
    samples = np.arange(w * h * 3).reshape(h, w * 3) % 7
    path = str(tmp_path / "key.png")
    write_png(path, _pack_rows(samples, 8), w, h, 2, 8, filters=[1], trns=[0, 3, 0, 4, 0, 5])   # the colour (3,4,5) is transparent
    got = sr.read_image(path)
    rgb = samples.reshape(h, w, 3)
    assert np.array_equal(got[..., 3] == 0, (rgb == [3, 4, 5]).all(axis=-1)) and (got[..., 3] == 0).any()
    with pytest.raises(sr.SoftRasterError, match="cannot open"):
        sr.read_image(str(tmp_path / "missing.png"))
    bad = tmp_path / "bad.png"
    data = bytearray(open(path, "rb").read())
    data[40] ^= 0xFF
    bad.write_bytes(bytes(data))
    with pytest.raises(sr.SoftRasterError, match="checksum|inflate"):
        sr.read_image(str(bad))
    jpg = tmp_path / "x.jpg"
    jpg.write_bytes(b"\xff\xd8\xff\xe0" + bytes(32))
    with pytest.raises(sr.SoftRasterError, match="JPEG"):
        sr.read_image(str(jpg))


@pytest.mark.parametrize("name,lo,hi", [("rustediron2_metallic.png", 0, 254), ("rustediron2_roughness.png", 72, 255)])
def test_reference_textures_decode_like_pillow(sr, name, lo, hi):
    """The two maps the reference ships (2048 x 2048, 8-bit gray), read where they lie: R = G = B = gray,
    A = 255 (src/texture.cpp:278-279), and the value ranges the reference's material has."""
    path = os.path.join(REF_TEXTURES, name)
    if not os.path.exists(path):
        pytest.skip("the reference checkout is not present on this machine")
    from PIL import Image
    got = sr.read_image(path)
    exp = np.asarray(Image.open(path).convert("RGBA"))
    assert got.shape == (2048, 2048, 4) and np.array_equal(got, exp)
    assert int(got[..., 0].min()) == lo and int(got[..., 0].max()) == hi and (got[..., 3] == 255).all()
    assert np.array_equal(got[..., 0], got[..., 1]) and np.array_equal(got[..., 0], got[..., 2])


# ---------------------------------------------------------------------------------------------- glTF
PIXEL_PNG = base64.b64encode(b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", 1, 1, 8, 2, 0, 0, 0)) +
                             _chunk(b"IDAT", zlib.compress(b"\x00\x10\x20\x30")) + _chunk(b"IEND", b"")).decode()


def make_gltf(meshes, embed="data", material=True, embedded_texture=True, interleave=False, short_indices=False, modes=None):
    """meshes: list of (verts (n,8) with the v the READER must return, indices).  Returns (json dict, bin bytes)."""
    blob = bytearray()
    views, accessors, gl_meshes, nodes = [], [], [], []

    def add_view(data, stride=None):
        while len(blob) % 4:
            blob.append(0)
        views.append({"buffer": 0, "byteOffset": len(blob), "byteLength": len(data), **({"byteStride": stride} if stride else {})})
        blob.extend(data)
        return len(views) - 1

    def add_accessor(view, ctype, count, kind, offset=0, normalized=False):
        accessors.append({"bufferView": view, "componentType": ctype, "count": count, "type": kind, "byteOffset": offset,
                          **({"normalized": True} if normalized else {})})
        return len(accessors) - 1

    for k, (verts, idx) in enumerate(meshes):
        n = verts.shape[0]
        file_uv = verts[:, 6:8].copy()
        file_uv[:, 1] = 1.0 - file_uv[:, 1]     # the reader flips v back (aiProcess_FlipUVs)
        if interleave:
            inter = np.concatenate([verts[:, 0:3], verts[:, 3:6], file_uv], axis=1).astype("<f4")
            v = add_view(inter.tobytes(), stride=32)
            pos, nrm, uv = add_accessor(v, 5126, n, "VEC3", 0), add_accessor(v, 5126, n, "VEC3", 12), add_accessor(v, 5126, n, "VEC2", 24)
        else:
            pos = add_accessor(add_view(verts[:, 0:3].astype("<f4").tobytes()), 5126, n, "VEC3")
            nrm = add_accessor(add_view(verts[:, 3:6].astype("<f4").tobytes()), 5126, n, "VEC3")
            uv = add_accessor(add_view(file_uv.astype("<f4").tobytes()), 5126, n, "VEC2")
        prim = {"attributes": {"POSITION": pos, "NORMAL": nrm, "TEXCOORD_0": uv}}
        if idx is not None:
            if short_indices:
                prim["indices"] = add_accessor(add_view(idx.astype("<u2").tobytes()), 5123, idx.size, "SCALAR")
            else:
                prim["indices"] = add_accessor(add_view(idx.astype("<u4").tobytes()), 5125, idx.size, "SCALAR")
        if material:
            prim["material"] = 0
        if modes:
            prim["mode"] = modes[k]
        gl_meshes.append({"primitives": [prim]})
    # node tree: node 0 is the root with the LAST mesh, its children carry the others in order -> depth-first order is last, 0, 1, ...
    nodes.append({"mesh": len(meshes) - 1, "children": list(range(1, len(meshes))), "translation": [5, 5, 5]})
    for k in range(len(meshes) - 1):
        nodes.append({"mesh": k, "scale": [2, 2, 2]})
    image = {"uri": "data:image/png;base64," + PIXEL_PNG} if embedded_texture else {"uri": "albedo.png"}
    doc = {"asset": {"version": "2.0"}, "scene": 0, "scenes": [{"nodes": [0]}], "nodes": nodes, "meshes": gl_meshes,
           "accessors": accessors, "bufferViews": views, "images": [image], "textures": [{"source": 0}],
           "materials": [{"pbrMetallicRoughness": {"baseColorTexture": {"index": 0}}, "name": "café \"quoted\""}],
           "buffers": [{"byteLength": len(blob)}]}
    return doc, bytes(blob)


def write_gltf(tmp_path, name, doc, blob, embed):
    path = str(tmp_path / name)
    if embed == "glb":
        js = json.dumps(doc).encode()
        js += b" " * (-len(js) % 4)
        bn = blob + bytes(-len(blob) % 4)
        with open(path, "wb") as f:
            f.write(b"glTF" + struct.pack("<II", 2, 12 + 8 + len(js) + 8 + len(bn)) + struct.pack("<II", len(js), 0x4E4F534A) + js +
                    struct.pack("<II", len(bn), 0x004E4942) + bn)
        return path
    if embed == "data":
        doc["buffers"][0]["uri"] = "data:application/octet-stream;base64," + base64.b64encode(blob).decode()
    else:
        doc["buffers"][0]["uri"] = name + ".bin"
        with open(path + ".bin", "wb") as f:
            f.write(blob)
    with open(path, "w") as f:
        json.dump(doc, f, indent=1)
    return path


def _two_meshes(sr):
    a = sr.scene.uv_sphere(9, 6)
    b = sr.scene.screen_quad(z=0.25, uv_scale=3.0)
    return [(a[0], a[1]), (b[0], b[1])]


@pytest.mark.parametrize("embed,interleave,short", [("data", False, False), ("file", True, True), ("glb", True, False)])
def test_gltf_round_trip(sr, tmp_path, embed, interleave, short):
    meshes = _two_meshes(sr)
    doc, blob = make_gltf(meshes, interleave=interleave, short_indices=short)
    path = write_gltf(tmp_path, "m.glb" if embed == "glb" else "m.gltf", doc, blob, embed)
    got = sr.read_model(path)
    # depth-first from the root: the root node holds the last mesh, its children the others in order; transforms are ignored
    order = [len(meshes) - 1] + list(range(len(meshes) - 1))
    assert len(got) == len(meshes)
    for (verts, idx), k in zip(got, order):
        assert np.array_equal(idx, meshes[k][1])
        assert np.array_equal(verts[:, :7], meshes[k][0][:, :7])
        assert np.allclose(verts[:, 7], meshes[k][0][:, 7], atol=1e-6)   # 1 - (1 - v) in float32


def test_gltf_strip_fan_and_unindexed(sr, tmp_path):
    quad = sr.scene.screen_quad()[0]
    strip = (quad[[0, 1, 3, 2]], np.array([0, 1, 2, 3], np.uint32))         # triangle strip over the quad
    fan = (quad, np.array([0, 1, 2, 3], np.uint32))                          # fan around vertex 0
    soup = (np.concatenate([quad[[0, 1, 2]], quad[[0, 2, 3]]]), None)        # no index accessor
    doc, blob = make_gltf([strip, fan, soup], modes=[5, 6, 4])
    got = sr.read_model(write_gltf(tmp_path, "p.gltf", doc, blob, "data"))
    by_mesh = {2: got[0], 0: got[1], 1: got[2]}
    assert by_mesh[0][1].tolist() == [0, 1, 2, 2, 1, 3]     # second strip triangle keeps the winding
    assert by_mesh[1][1].tolist() == [0, 1, 2, 0, 2, 3]
    assert by_mesh[2][1].tolist() == [0, 1, 2, 3, 4, 5]


def test_gltf_errors_mirror_the_reference(sr, tmp_path):
    meshes = _two_meshes(sr)
    doc, blob = make_gltf(meshes, material=False)
    with pytest.raises(sr.SoftRasterError, match="No material!"):                       # src/model.cpp:52-54
        sr.read_model(write_gltf(tmp_path, "a.gltf", doc, blob, "data"))
    doc, blob = make_gltf(meshes, embedded_texture=False)
    with pytest.raises(sr.SoftRasterError, match="Cannot load diffuse texture"):         # src/model.cpp:47-49
        sr.read_model(write_gltf(tmp_path, "b.gltf", doc, blob, "data"))
    doc, blob = make_gltf(meshes)
    del doc["scenes"]
    with pytest.raises(sr.SoftRasterError, match="cannot read scene"):                   # src/model.cpp:88-91
        sr.read_model(write_gltf(tmp_path, "c.gltf", doc, blob, "data"))
    doc, blob = make_gltf(meshes)
    doc["accessors"][0]["count"] = 10 ** 6
    with pytest.raises(sr.SoftRasterError, match="past its buffer"):
        sr.read_model(write_gltf(tmp_path, "d.gltf", doc, blob, "data"))
    (tmp_path / "e.gltf").write_text("{ \"asset\": [1, 2,, ] }")
    with pytest.raises(sr.SoftRasterError, match="JSON"):
        sr.read_model(str(tmp_path / "e.gltf"))
    doc, blob = make_gltf(meshes)
    doc["nodes"][1]["children"] = [0]   # a cycle
    with pytest.raises(sr.SoftRasterError, match="too deep"):
        sr.read_model(write_gltf(tmp_path, "f.gltf", doc, blob, "data"))
