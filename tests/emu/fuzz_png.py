"""TEST INFRASTRUCTURE ONLY: structure-aware PNG mutations (header fields, palette and transparency lengths, scan-line bytes; checksums and compression kept valid) -- run fuzz_files.py first, it writes the seed files, fed to the readers of vulkan-renderer_amd/csrc/sr_assets.cpp built with
AddressSanitizer and UndefinedBehaviorSanitizer (tests/emu/asan_files.cpp).  The readers may refuse a file; any sanitizer report
fails the run.    python tests/emu/fuzz_png.py [seed [rounds of 100 files]]"""
import os, sys, subprocess, random, struct, zlib
from pathlib import Path
out = Path('/tmp/file_fuzz'); exe = str(out / 'asan_files')
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 7)
seeds = sorted(out.glob('seed_*.png'))
def chunks(data):
    pos = 8; res = []
    while pos + 12 <= len(data):
        n = struct.unpack('>I', data[pos:pos+4])[0]; res.append((data[pos+4:pos+8], bytearray(data[pos+8:pos+8+n]))); pos += 12 + n
    return res
def build(ch):
    b = bytearray(b'\x89PNG\r\n\x1a\n')
    for t, body in ch: b += struct.pack('>I', len(body)) + t + body + struct.pack('>I', zlib.crc32(t + bytes(body)) & 0xFFFFFFFF)
    return bytes(b)
total = 0; acc = 0
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    batch = []
    for k in range(100):
        ch = chunks(seeds[rng.randrange(len(seeds))].read_bytes())
        idat = b''.join(bytes(b) for t, b in ch if t == b'IDAT'); raw = bytearray(zlib.decompress(idat))
        for _ in range(rng.randrange(1, 4)):
            m = rng.randrange(7)
            ih = next(b for t, b in ch if t == b'IHDR')
            if m == 0: ih[0:4] = struct.pack('>I', rng.choice([0, 1, 2, 12, 13, 14, 100, 32768, 32769, 0x7FFFFFFF]))
            elif m == 1: ih[4:8] = struct.pack('>I', rng.choice([0, 1, 6, 7, 8, 100, 32768, 0xFFFFFFFF]))
            elif m == 2: ih[8] = rng.choice([0, 1, 2, 3, 4, 8, 16, 32])
            elif m == 3: ih[9] = rng.choice([0, 1, 2, 3, 4, 5, 6, 7])
            elif m == 4:
                for _ in range(rng.randrange(1, 6)): raw[rng.randrange(len(raw))] = rng.randrange(256)
            elif m == 5:
                raw = raw[:rng.randrange(0, len(raw) + 1)] if rng.randrange(2) else raw + bytes(rng.randrange(256) for _ in range(rng.randrange(1, 200)))
                if not raw: raw = bytearray(b'\x00')
            else:
                for i, (t, b) in enumerate(ch):
                    if t in (b'PLTE', b'tRNS') and rng.randrange(2): ch[i] = (t, bytearray(b[:rng.randrange(0, len(b) + 1)]))
        new = []; done = False
        for t, b in ch:
            if t == b'IDAT':
                if not done: new.append((t, bytearray(zlib.compress(bytes(raw))))); done = True
            else: new.append((t, b))
        p = out / f'p{k}.png'; p.write_bytes(build(new)); batch.append(str(p))
    r = subprocess.run([exe] + batch, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, ASAN_OPTIONS='verify_asan_link_order=0:detect_leaks=0:allocator_may_return_null=1'))
    total += 100
    if r.returncode != 0:
        print('SANITIZER REPORT in round', rnd); print(r.stdout[-3000:]); sys.exit(1)
    acc += int(r.stdout.split()[1])
print('ok', total, 'structured PNG mutations,', acc, 'accepted')
