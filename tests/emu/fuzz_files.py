"""TEST INFRASTRUCTURE ONLY: byte-level mutations (bit flips, truncations, 32-bit overwrites, number swaps in the JSON, insertions) of valid PNG, JPEG, .gltf and .glb files, fed to the readers of vulkan-renderer_amd/csrc/sr_assets.cpp and sr_jpeg.cpp built with
AddressSanitizer and UndefinedBehaviorSanitizer (tests/emu/asan_files.cpp).  The readers may refuse a file; any sanitizer report
fails the run.    python tests/emu/fuzz_files.py [seed [rounds of 100 files]]"""
import os, sys, subprocess, json, random, struct
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'vulkan-renderer_amd'), os.path.join(ROOT, 'tests')]
import numpy as np
import test_asset_files as F
import softraster as sr
from pathlib import Path
out = Path('/tmp/file_fuzz'); out.mkdir(exist_ok=True)
exe = str(out / 'asan_files')
subprocess.check_call(['g++','-std=c++17','-O1','-g','-fsanitize=address,undefined','-fno-sanitize-recover=all','-static-libasan','-static-libubsan',
  '-I',os.path.join(ROOT, 'vulkan-renderer_amd', 'csrc'),os.path.join(ROOT, 'tests', 'emu', 'asan_files.cpp'),os.path.join(ROOT, 'vulkan-renderer_amd', 'csrc', 'sr_assets.cpp'),os.path.join(ROOT, 'vulkan-renderer_amd', 'csrc', 'sr_jpeg.cpp'),'-lz','-o',exe])
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
seeds = []
# valid seed files
for (ct, d) in [(0,1),(0,8),(0,16),(2,8),(3,2),(3,8),(4,8),(6,8),(6,16)]:
    w,h = 13,7
    ch = {0:1,2:3,3:1,4:2,6:4}[ct]
    nr = np.random.default_rng(ct*10+d)
    samples = nr.integers(0, 1<<d, size=(h, w*ch))
    p = out / f'seed_{ct}_{d}.png'
    F.write_png(str(p), F._pack_rows(samples, d), w, h, ct, d, filters=[0,1,2,3,4], plte=list(range(48)) if ct==3 else None, trns=[1,2,3] if ct==3 else None)
    seeds.append(p)
    if d in (2, 8):   # the same kind once more with Adam7 interlacing
        p = out / f'seed_{ct}_{d}_i.png'
        F.write_png(str(p), None, w, h, ct, d, filters=[4,1,3], plte=list(range(48)) if ct==3 else None, trns=[1,2,3] if ct==3 else None, interlaced_samples=samples)
        seeds.append(p)
for n, kw in enumerate([dict(subsampling=2), dict(subsampling=1, progressive=True), dict(subsampling=0, progressive=True, optimize=True), dict(restart_marker_blocks=2)]):
    p = out / f'seed_{n}.jpg'
    F._test_picture(37, 21, n, gray=(n == 3)).save(str(p), 'JPEG', quality=85, **kw)
    seeds.append(p)
for n, spec in enumerate([F.JPEG_STRUCTURES[i] for i in (1, 3, 8, 9, 10)]):
    p = out / f'seed_s{n}.jpg'
    F.write_baseline_jpeg(str(p), 29, 19, seed=n, **spec)
    seeds.append(p)
meshes = [sr.scene.uv_sphere(5,4), sr.scene.screen_quad()]
for embed, name in (('data','a.gltf'),('glb','b.glb')):
    doc, blob = F.make_gltf(meshes, interleave=(embed=='glb'), short_indices=(embed=='data'))
    seeds.append(Path(F.write_gltf(out, name, doc, blob, embed)))
total = 0
for rnd in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    batch = []
    for k in range(100):
        src = rng.choice(seeds); data = bytearray(src.read_bytes())
        mode = rng.randrange(5)
        if mode == 0:
            for _ in range(rng.randrange(1, 8)): data[rng.randrange(len(data))] ^= 1 << rng.randrange(8)
        elif mode == 1:
            data = data[:rng.randrange(1, len(data))]
        elif mode == 2:
            i = rng.randrange(len(data)); data[i:i+4] = struct.pack('>I', rng.choice([0, 1, 0x7FFFFFFF, 0xFFFFFFFF, rng.randrange(1<<32)]))
        elif mode == 3 and src.suffix == '.gltf':
            txt = data.decode()
            import re
            nums = list(re.finditer(r'\d+', txt))
            m = rng.choice(nums); txt = txt[:m.start()] + str(rng.choice([0, 1, 65535, 2**31, 2**32-1, 10**12, -5])) + txt[m.end():]
            data = bytearray(txt.encode())
        else:
            i = rng.randrange(len(data)); data[i:i] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 16)))
        p = out / f'm{k}{src.suffix}'
        p.write_bytes(bytes(data)); batch.append(str(p))
    r = subprocess.run([exe] + batch, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, env=dict(os.environ, ASAN_OPTIONS='verify_asan_link_order=0:detect_leaks=0:allocator_may_return_null=1'))
    total += len(batch)
    if r.returncode != 0:
        print('SANITIZER REPORT in round', rnd); print(r.stdout[-3000:]); sys.exit(1)
print('ok', total, 'mutated files;', r.stdout.strip())
