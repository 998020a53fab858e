"""File readers behind create_texture_from_file and Model::load (host-only entry points of the C ABI;
no GPU needed).  JPEG: files written by libjpeg-turbo (Pillow) and by a small encoder here, decoded by the
library's libjpeg flavour (bit for bit what libjpeg-turbo gives) and its default stb_image flavour; and
the JPEG the reference ships.  PNG: files written here in every colour type / bit depth / filter, decoded by the
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


def _filter_rows(rows, bpp, filters):
    """Packed scan lines -> filtered stream (filter type byte + residues per line)."""
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
    return raw


ADAM7 = [(0, 0, 8, 8), (4, 0, 8, 8), (0, 4, 4, 8), (2, 0, 4, 4), (0, 2, 2, 4), (1, 0, 2, 2), (0, 1, 1, 2)]   # x0, y0, dx, dy


def write_png(path, rows, width, height, ctype, depth, filters, plte=None, trns=None, interlaced_samples=None):
    """rows: list of bytes objects (already packed scan lines); filters: one filter type per row.
    interlaced_samples: instead of rows, the (height, width*channels) array of raw samples, written as the seven
    passes of Adam7 interlacing."""
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    bpp = max(channels * depth // 8, 1)
    if interlaced_samples is None:
        raw = _filter_rows(rows, bpp, filters)
    else:
        raw = bytearray()
        px = np.asarray(interlaced_samples).reshape(height, width, channels)
        for x0, y0, dx, dy in ADAM7:
            sub = px[y0::dy, x0::dx]
            if sub.shape[0] and sub.shape[1]:
                raw += _filter_rows(_pack_rows(sub.reshape(sub.shape[0], -1), depth), bpp, filters)
    data = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", width, height, depth, ctype, 0, 0, 0 if interlaced_samples is None else 1))
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


@pytest.mark.parametrize("ctype,depth", [(0, 1), (0, 4), (0, 16), (2, 8), (3, 2), (3, 8), (4, 8), (6, 8), (6, 16)])
def test_interlaced_png_matches_the_plain_one(sr, tmp_path, ctype, depth):
    """Adam7: the same samples written plain and interlaced decode to the same pixels (and to what Pillow reads), for
    sizes down to 1 x 1, where most of the seven passes are empty."""
    from PIL import Image
    rng = np.random.default_rng(ctype * 100 + depth + 7)
    channels = {0: 1, 2: 3, 3: 1, 4: 2, 6: 4}[ctype]
    for w, h in [(37, 23), (1, 1), (3, 2), (5, 9), (8, 8), (2, 5)]:
        samples = rng.integers(0, 1 << depth, size=(h, w * channels))
        plte = rng.integers(0, 256, size=3 * (1 << depth)).tolist() if ctype == 3 else None
        trns = rng.integers(0, 256, size=(1 << depth) // 2 + 1).tolist() if ctype == 3 else None
        plain, laced = str(tmp_path / "plain.png"), str(tmp_path / "laced.png")
        write_png(plain, _pack_rows(samples, depth), w, h, ctype, depth, filters=[4, 0, 3], plte=plte, trns=trns)
        write_png(laced, None, w, h, ctype, depth, filters=[1, 4, 2, 3, 0], plte=plte, trns=trns, interlaced_samples=samples)
        got = sr.read_image(laced)
        assert got.shape == (h, w, 4) and np.array_equal(got, sr.read_image(plain)), (w, h)
        if depth != 16:
            assert np.array_equal(got, np.asarray(Image.open(laced).convert("RGBA"))), (w, h)


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


# ---------------------------------------------------------------------------------------------- JPEG
ZIGZAG = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28, 35, 42, 49, 56,
          57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _libjpeg(path):
    """What libjpeg-turbo (through Pillow) makes of the file, as RGBA."""
    from PIL import Image
    return np.asarray(Image.open(path).convert("RGBA"))


def _test_picture(w, h, seed, gray=False):
    from PIL import Image
    y, x = np.mgrid[0:h, 0:w]
    a = np.stack([(x * 7 + y * 3) & 255, ((x ^ y) * 5) & 255, (x * y) & 255], -1)
    a = (a // 2 + np.random.default_rng(seed).integers(0, 128, a.shape)).astype(np.uint8)   # structure plus noise: every coefficient is busy
    im = Image.fromarray(a, "RGB")
    return im.convert("L") if gray else im


JPEG_SIZES = [(64, 48), (17, 13), (1, 1), (2, 2), (3, 5), (5, 3), (33, 31), (100, 75), (16, 16), (15, 17)]


@pytest.mark.parametrize("progressive", [False, True])
@pytest.mark.parametrize("layout", ["gray", "444", "422", "420"])
def test_jpeg_matches_libjpeg(sr, tmp_path, monkeypatch, layout, progressive):
    """Files written by libjpeg-turbo (Pillow): baseline and progressive (spectral selection and successive
    approximation, end-of-band runs), 4:4:4 / 4:2:2 / 4:2:0 and gray, three qualities, restart markers, optimised
    Huffman tables, sizes that are not whole MCUs.  The libjpeg flavour of the reader must give the same bytes as
    libjpeg-turbo; the default flavour (stb_image's arithmetic) must stay within a few code values of it."""
    for n, (w, h) in enumerate(JPEG_SIZES):
        for quality in (30, 90, 100):
            for extra in ({}, {"restart_marker_blocks": 3}, {"optimize": True}):
                path = str(tmp_path / "t.jpg")
                kw = dict(quality=quality, progressive=progressive, **extra)
                if layout != "gray":
                    kw["subsampling"] = {"444": 0, "422": 1, "420": 2}[layout]
                _test_picture(w, h, n, gray=layout == "gray").save(path, "JPEG", **kw)
                exp = _libjpeg(path)
                monkeypatch.setenv("SR_JPEG_DECODER", "libjpeg")
                got = sr.read_image(path)
                assert got.shape == exp.shape and np.array_equal(got, exp), (w, h, quality, extra)
                monkeypatch.delenv("SR_JPEG_DECODER")
                stb = sr.read_image(path)
                d = np.abs(stb.astype(int) - exp.astype(int)).max(axis=-1)
                dw = (w + 1) // 2
                if layout == "422":
                    d[:, 2 * dw - 2] = 0    # stb_image weighs the last interpolated column the other way round
                if layout in ("gray", "444") or dw > 2:   # (libjpeg repeats samples instead of filtering when a chroma row has one or two)
                    assert d.max() <= 3, (w, h, quality, extra, int(d.max()))


def _huffman_table(symbols, skewed):
    """A valid code for the symbols: all eight bits long, or short codes for the first few and 16-bit codes
    for the rest (the decoder's slow path).  Returns (counts[16], symbols in code order, {symbol: (code, length)})."""
    if not skewed or len(symbols) < 8:
        n_bits = 4 if len(symbols) < 15 else 8
        lengths = [n_bits] * len(symbols)
    else:
        lengths = ([3] + [4] * 2 + [5] * 4 + [6] * 8 + [7] * 16 + [8] * 32)[:len(symbols)]
        lengths += [16] * (len(symbols) - len(lengths))
    counts, codes, code, k = [0] * 16, {}, 0, 0
    for length in range(1, 17):
        for _ in range(lengths.count(length)):
            codes[symbols[k]] = (code, length)
            code += 1
            k += 1
        counts[length - 1] = lengths.count(length)
        code <<= 1
    return counts, symbols, codes


class _Bits:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 255
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n & 7:
            self.put((1 << (8 - (self.n & 7))) - 1, 8 - (self.n & 7))


def write_baseline_jpeg(path, width, height, comps, seed, scans=None, restart=0, skewed=False, wide_quant=False, jfif=True,
                        adobe_transform=None, fill_bytes=False):
    """A baseline JPEG made of random quantised coefficients (no picture behind it: the decoders are compared on
    whatever it decodes to).  comps: [(id, h, v)]; scans: lists of component indices, default one interleaved scan."""
    rng = np.random.default_rng(seed)
    hmax, vmax = max(c[1] for c in comps), max(c[2] for c in comps)
    mcux, mcuy = -(-width // (8 * hmax)), -(-height // (8 * vmax))
    seg = lambda m, body: bytes([0xFF, m]) + struct.pack(">H", len(body) + 2) + body
    out = bytearray(b"\xff\xd8")
    if jfif:
        out += seg(0xE0, b"JFIF\0\x01\x01\x00\x00\x01\x00\x01\x00\x00")
    if adobe_transform is not None:
        out += seg(0xEE, b"Adobe\0\x64\0\0\0\0" + bytes([adobe_transform]))
    quant = rng.integers(1, 12, size=(2, 64))
    for t in range(2):
        out += seg(0xDB, bytes([0x10 | t]) + b"".join(struct.pack(">H", int(q)) for q in quant[t]) if wide_quant else bytes([t]) + bytes(int(q) for q in quant[t]))
    out += seg(0xC0, struct.pack(">BHHB", 8, height, width, len(comps)) + b"".join(bytes([cid, (h << 4) | v, min(i, 1)]) for i, (cid, h, v) in enumerate(comps)))
    dc_counts, dc_syms, dc_codes = _huffman_table(list(range(12)), False)
    ac_syms = [0x00, 0xF0] + [(r << 4) | s for s in range(1, 11) for r in range(16)]
    ac_counts, ac_syms, ac_codes = _huffman_table(ac_syms, skewed)
    out += seg(0xC4, bytes([0x00]) + bytes(dc_counts) + bytes(dc_syms) + bytes([0x10]) + bytes(ac_counts) + bytes(ac_syms))
    if restart:
        out += seg(0xDD, struct.pack(">H", restart))
    # coefficients in zig-zag order for every block the frame allocates; sparse, small
    coefs = []
    for (cid, h, v) in comps:
        c = rng.integers(-20, 21, size=(mcuy * v, mcux * h, 64)) * (rng.random((mcuy * v, mcux * h, 64)) < 0.15)
        c[..., 0] = rng.integers(-40, 41, size=c.shape[:2])
        c[..., 63] = rng.integers(-2, 3, size=c.shape[:2])     # the last coefficient too, now and then after a long zero run
        c[::3, ::2, 1:] = 0                                      # and blocks with a DC term only
        coefs.append(c)

    def magnitude(bits, v):
        s = int(abs(v)).bit_length()
        bits.put(v if v >= 0 else v + (1 << s) - 1, s)

    for scan in (scans or [list(range(len(comps)))]):
        if fill_bytes:
            out += b"\xff\xff\xff"
        out += seg(0xDA, bytes([len(scan)]) + b"".join(bytes([comps[i][0], 0x00]) for i in scan) + bytes([0, 63, 0]))
        bits, pred, count, rst = _Bits(), [0] * len(comps), 0, 0
        if len(scan) == 1:
            cid, h, v = comps[scan[0]]
            nx, ny = -(-(-(-width * h // hmax)) // 8), -(-(-(-height * v // vmax)) // 8)
        else:
            nx, ny = mcux, mcuy
        for my in range(ny):
            for mx in range(nx):
                if restart and count == restart:
                    bits.flush()
                    bits.out += bytes([0xFF, 0xD0 + rst])
                    rst, count, pred = (rst + 1) & 7, 0, [0] * len(comps)
                count += 1
                for i in scan:
                    cid, h, v = comps[i]
                    for by in range(v if len(scan) > 1 else 1):
                        for bx in range(h if len(scan) > 1 else 1):
                            blk = coefs[i][my * (v if len(scan) > 1 else 1) + by, mx * (h if len(scan) > 1 else 1) + bx]
                            diff = int(blk[0]) - pred[i]
                            pred[i] = int(blk[0])
                            bits.put(*dc_codes[int(abs(diff)).bit_length()])
                            magnitude(bits, diff)
                            run = 0
                            last = max([k for k in range(1, 64) if blk[k]], default=0)
                            for k in range(1, last + 1):
                                if blk[k] == 0:
                                    run += 1
                                    continue
                                while run > 15:
                                    bits.put(*ac_codes[0xF0])
                                    run -= 16
                                bits.put(*ac_codes[(run << 4) | int(abs(blk[k])).bit_length()])
                                magnitude(bits, int(blk[k]))
                                run = 0
                            if last < 63:
                                bits.put(*ac_codes[0x00])
        bits.flush()
        out += bits.out
    out += b"\xff\xd9"
    with open(path, "wb") as f:
        f.write(bytes(out))


JPEG_STRUCTURES = [
    dict(comps=[(1, 1, 1), (2, 1, 1), (3, 1, 1)]),
    dict(comps=[(1, 1, 2), (2, 1, 1), (3, 1, 1)]),                                # 4:4:0, the filter that runs down only
    dict(comps=[(1, 4, 1), (2, 1, 1), (3, 1, 1)]),                                # 4:1:1, samples repeated four times
    dict(comps=[(1, 4, 1), (2, 2, 1), (3, 1, 1)]),                                # a different ratio per component
    dict(comps=[(1, 2, 2), (2, 2, 1), (3, 1, 2)]),
    dict(comps=[(1, 3, 1), (2, 1, 1), (3, 1, 1)]),
    dict(comps=[(1, 1, 3), (2, 1, 1), (3, 1, 1)]),
    dict(comps=[(1, 2, 2), (2, 1, 1), (3, 1, 1)], scans=[[0], [1], [2]]),         # one scan per component
    dict(comps=[(1, 2, 1), (2, 1, 1), (3, 1, 1)], scans=[[0], [1, 2]], restart=2),
    dict(comps=[(1, 2, 2), (2, 1, 1), (3, 1, 1)], restart=1, skewed=True),
    dict(comps=[(1, 2, 2), (2, 1, 1), (3, 1, 1)], restart=5, wide_quant=True, fill_bytes=True),
    dict(comps=[(7, 2, 2)]),                                                      # gray whose sampling factors mean nothing
    dict(comps=[(1, 1, 1)], restart=3, skewed=True),
    dict(comps=[(82, 1, 1), (71, 1, 1), (66, 1, 1)], jfif=False),                 # components named R, G, B: no colour transform
    dict(comps=[(1, 2, 1), (2, 1, 1), (3, 1, 1)], jfif=False, adobe_transform=0),
    dict(comps=[(1, 2, 1), (2, 1, 1), (3, 1, 1)], jfif=False, adobe_transform=1),
]


@pytest.mark.parametrize("case", range(len(JPEG_STRUCTURES)))
def test_jpeg_structures_libjpeg_turbo_does_not_write(sr, tmp_path, monkeypatch, case):
    """Sampling ratios, scan layouts, restart intervals, 16-bit quantisation tables, 16-bit Huffman codes, fill
    bytes and colour-space markers that Pillow's encoder never produces, written by the small encoder above;
    libjpeg-turbo (through Pillow) still decodes them and is the reference."""
    spec = JPEG_STRUCTURES[case]
    for n, (w, h) in enumerate([(40, 24), (37, 29), (7, 9), (65, 33)]):
        path = str(tmp_path / "s.jpg")
        write_baseline_jpeg(path, w, h, seed=case * 10 + n, **spec)
        exp = _libjpeg(path)
        monkeypatch.setenv("SR_JPEG_DECODER", "libjpeg")
        got = sr.read_image(path)
        assert got.shape == exp.shape and np.array_equal(got, exp), (spec, w, h)
        monkeypatch.delenv("SR_JPEG_DECODER")
        stb = sr.read_image(path)
        assert stb.shape == exp.shape and (stb[..., 3] == 255).all()
        if all(c[1:] == (1, 1) for c in spec["comps"]) or len(spec["comps"]) == 1:   # no upsampling: only the IDCT and the colour arithmetic differ
            assert np.abs(stb.astype(int) - exp.astype(int)).max() <= 3


def test_jpeg_kinds_that_are_refused(sr, tmp_path):
    from PIL import Image
    path = str(tmp_path / "cmyk.jpg")
    Image.new("CMYK", (8, 8), (10, 20, 30, 40)).save(path, "JPEG")
    with pytest.raises(sr.SoftRasterError, match="three-component"):
        sr.read_image(path)
    good = str(tmp_path / "good.jpg")
    _test_picture(24, 24, 0).save(good, "JPEG")
    data = open(good, "rb").read()
    sof = data.index(b"\xff\xc0")
    for marker, what in ((0xC9, "arithmetic"), (0xC3, "lossless")):
        bad = tmp_path / "bad.jpg"
        bad.write_bytes(data[:sof + 1] + bytes([marker]) + data[sof + 2:])
        with pytest.raises(sr.SoftRasterError, match=what):
            sr.read_image(str(bad))
    bad = tmp_path / "deep.jpg"
    bad.write_bytes(data[:sof + 4] + b"\x0c" + data[sof + 5:])      # 12 bits per sample
    with pytest.raises(sr.SoftRasterError, match="8-bit"):
        sr.read_image(str(bad))
    bad.write_bytes(data[:sof + 5] + b"\x7f\xff\x7f\xff" + data[sof + 9:])   # a size the data cannot hold
    with pytest.raises(sr.SoftRasterError, match="too short"):
        sr.read_image(str(bad))
    cut = tmp_path / "cut.jpg"
    sos = data.index(b"\xff\xda")
    cut.write_bytes(data[:sos + (len(data) - sos) * 2 // 3])        # the tail is missing: what is there decodes
    got = sr.read_image(str(cut))
    assert got.shape == (24, 24, 4)
    assert np.array_equal(got[:8], sr.read_image(good)[:8])


def test_reference_albedo_texture(sr, monkeypatch):
    """data/textures/chalet.jpg, the one JPEG the reference ships (4096 x 4096, baseline, 4:2:0), read where it
    lies: the libjpeg flavour gives libjpeg-turbo's bytes, the default flavour stays within 3 code values of
    them with more than 95 % of the bytes equal."""
    path = "/root/reference/data/textures/chalet.jpg"
    if not os.path.exists(path):
        pytest.skip("the reference checkout is not present on this machine")
    exp = _libjpeg(path)
    monkeypatch.setenv("SR_JPEG_DECODER", "libjpeg")
    got = sr.read_image(path)
    assert got.shape == (4096, 4096, 4) and np.array_equal(got, exp)
    monkeypatch.delenv("SR_JPEG_DECODER")
    stb = sr.read_image(path)
    d = np.abs(stb.astype(np.int16) - exp.astype(np.int16))
    assert d.max() <= 3 and (d == 0).mean() > 0.95 and (stb[..., 3] == 255).all()
    assert abs(float(stb[..., :3].mean()) - 88.2) < 0.1


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
