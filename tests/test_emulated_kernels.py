"""The kernel sources, compiled for the host and executed work-item by work-item on the CPU (tests/emu),
must pass the same parity tests the GPU library passes -- with every device allocation fenced by
no-access pages, so an out-of-range load or store of any kernel fails here, before it can fault a GPU.

This is a check of the kernels' indexing, synchronisation and integer arithmetic.  It is not a way to
run the renderer without a GPU: the emulation library lives under tests/, is built from here only, and
nothing in the shipped package knows about it.
"""
import os
import subprocess
import sys


ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def start_gpu_suite_emulated(order="forward", select=None, what="tests", **extra_env):
    """Starts `pytest -m gpu` on the emulated kernels in a child process."""
    env = dict(os.environ, SR_TEST_EMULATED="1", HIPEMU_ORDER=order, **extra_env)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, what), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"]
    if select:
        cmd += ["-k", select]
    label = f"order={order} select={select} {extra_env}"
    return label, subprocess.Popen(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


def finish(run, timeout=1500):
    label, proc = run
    out, _ = proc.communicate(timeout=timeout)
    tail = "\n".join(out.splitlines()[-40:])
    assert proc.returncode == 0, f"emulated run ({label}) failed:\n{tail}"
    assert " passed" in tail and "failed" not in tail, tail
    return tail


def build_emulation_library():
    sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
    import build_emu
    build_emu.build()   # once, before child processes race to build it


# device-level knobs under which every frame must stay the same: the other tile size (64 x 64 at one sample, 16 x 16 at four), a tile reference list that overflows on
# every first frame, the staged resolve of set-up for small launches too, a scratch budget that splits batches into passes, chunk culling of small meshes
KNOBS = [dict(SR_TILE_LOG2="6", SR_FILL_BLOCKS="1"), dict(SR_REF_CAPACITY="16", SR_STAGED_MIN_BLOCKS="1"), dict(SR_SCRATCH_MIB="1", SR_TILE_LOG2="4", SR_CULL_MIN_CHUNKS="1")]
_child_runs = {}


def child_runs():
    """The child processes of the three long tests below, all started together the first time one of the tests asks:
    they keep the machine's cores busy side by side instead of test after test."""
    if _child_runs:
        return _child_runs
    build_emulation_library()
    subset = "soup or huge or golden or split or multi_view or errors or trailing or reused or far_plane"
    full = os.path.join("tests", "test_gpu_zz_full_size.py")
    # (the 64-view, 8K and 16K configurations of that file run at reduced size in the other suites: at full size they take minutes here)
    full_size = dict(what=full, select="not config and not largest", SR_EMULATE_FULL_SIZE="1", HIPEMU_FUZZ_ULP="1")
    _child_runs["full_size"] = [start_gpu_suite_emulated(**full_size)]
    _child_runs["default"] = [start_gpu_suite_emulated("forward"), start_gpu_suite_emulated("reverse", select=subset), start_gpu_suite_emulated("shuffle", select=subset)]
    _child_runs["knobs"] = [start_gpu_suite_emulated("forward", **k) for k in KNOBS]
    return _child_runs


def test_gpu_parity_suite_on_emulated_kernels():
    """Whole gpu-marked suite on the default pipeline, lanes scheduled in ascending order; a second and
    third run of the order-sensitive tests with the lanes of every workgroup in descending and in
    shuffled order between synchronisation points (a missing barrier or an order-dependent reservation
    shows up as a parity failure)."""
    tails = [finish(r) for r in child_runs()["default"]]
    # the two tests that need a real GPU process are the only ones allowed to be skipped
    assert "2 skipped" in tails[0], tails[0]


def test_device_knobs_give_the_same_frames():
    """Tile size, initial size of the tile reference list and the scratch budget are per-device settings (environment variables read by sr_device_create) that must not change a frame:
    the whole parity suite under each of them."""
    for r in child_runs()["knobs"]:
        finish(r)


def test_full_size_frame_on_emulated_kernels():
    """The 10 M-triangle 3840x2160 frame of the headline benchmark, executed on the CPU (about a
    minute), with every result of the hardware's approximate reciprocal / rsqrt / exp2 / log2 moved
    by up to one ulp: the whole frame against the oracle plus the whole-frame properties."""
    for r in child_runs()["full_size"]:
        finish(r)


def test_cpp_host_application_on_emulated_kernels(tmp_path):
    """The C++ mirror of the reference's host API (vulkan-renderer_amd/host), built against the emulation
    library: Model::load + create_texture_from_file + draw write the same frame as the Python host does
    from the same files."""
    import numpy as np
    build_emulation_library()
    import build_emu
    sys.path.insert(0, os.path.join(ROOT, "vulkan-renderer_amd"))
    import softraster as sr
    import test_asset_files as files
    app = str(tmp_path / "headless_app_emu")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-ffp-contract=off", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "vulkan-renderer_amd", "host", "headless_app.cpp"), "-o", app,
                           "-L", os.path.dirname(build_emu.LIB), "-lsoftraster_emu", "-Wl,-rpath," + os.path.dirname(build_emu.LIB)])
    sphere = sr.scene.uv_sphere(24, 16)
    quad = sr.scene.screen_quad(z=-0.3, half=1.2)
    doc, blob = files.make_gltf([quad, sphere])            # node order: sphere (root), then quad -> the quad's frame is the one kept
    gltf = files.write_gltf(tmp_path, "scene.gltf", doc, blob, "file")
    images = sr.scene.material_textures(16, seed=2)
    paths = []
    for k, im in enumerate(images):
        paths.append(str(tmp_path / f"map{k}.png"))
        files.write_png(paths[-1], files._pack_rows(im.reshape(16, 16 * 4).astype(np.int64), 8), 16, 16, 6, 8, filters=[4, 1])
    out = str(tmp_path / "frame.ppm")
    r = subprocess.run([app, "160", "120", "8", "8", out, "--model", gltf, "--textures", *paths], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    assert "mesh with %d indices" % sphere[1].size in r.stdout and "mesh with %d indices" % quad[1].size in r.stdout, r.stdout
    header, rest = open(out, "rb").read().split(b"\n255\n", 1)
    assert header == b"P6\n160 120"
    rgb = np.frombuffer(rest, np.uint8).reshape(120, 160, 3)
    # once more after four key presses (the reference's key_callback moves the camera, src/main.cpp:223-267)
    out_keys = str(tmp_path / "frame_keys.ppm")
    r = subprocess.run([app, "160", "120", "8", "8", out_keys, "--model", gltf, "--textures", *paths, "--keys", "WAAD"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                       text=True, timeout=300)
    assert r.returncode == 0, r.stdout
    camera_line = [l for l in r.stdout.splitlines() if l.startswith("camera ")][0].split()
    cam = sr.scene.OrbitCamera().keys("WAAD")
    assert np.allclose([float(x) for x in camera_line[1:4]], cam.camera_pos, rtol=0, atol=1e-6) and np.allclose([float(x) for x in camera_line[5:8]], cam.up, rtol=0, atol=1e-6)
    rgb_keys = np.frombuffer(open(out_keys, "rb").read().split(b"\n255\n", 1)[1], np.uint8).reshape(120, 160, 3)
    # the same frames from the Python host, in a child process that loads the emulation library
    script = f"""
import sys, numpy as np
sys.path[:0] = [{os.path.join(ROOT, 'vulkan-renderer_amd')!r}, {os.path.join(ROOT, 'tests')!r}]
import softraster as sr
sr.LIB_PATH = {build_emu.LIB!r}
dev = sr.Device(0)
model = sr.Model.load(dev, {gltf!r})
tex = [sr.create_texture_from_file(dev, p) for p in {paths!r}]
tgt = sr.RenderTarget(dev, 160, 120)
sr.draw(dev, tgt, model.meshes()[-1], sr.scene.default_camera_ubo(160, 120), tex)
np.save({str(tmp_path / 'py.npy')!r}, tgt.read_color(0))
sr.draw(dev, tgt, model.meshes()[-1], sr.scene.OrbitCamera().keys("WAAD").ubo(160, 120), tex)
np.save({str(tmp_path / 'py_keys.npy')!r}, tgt.read_color(0))
"""
    subprocess.run([sys.executable, "-c", script], check=True, timeout=300)
    bgra = np.load(str(tmp_path / "py.npy"))
    assert (rgb != 0).any()
    assert np.array_equal(rgb, bgra[..., [2, 1, 0]])
    # the moved camera: the two hosts' transcendental functions may differ in the last bit of the eye position
    bgra_keys = np.load(str(tmp_path / "py_keys.npy"))
    assert (rgb_keys != rgb).any()
    close = (np.abs(rgb_keys.astype(int) - bgra_keys[..., [2, 1, 0]].astype(int)).max(axis=-1) <= 2).mean()
    assert close >= 0.995, close


def test_scene_driver_under_address_and_ub_sanitizers():
    """tests/emu/asan_scenes.cpp: spheres and triangle soups through the C ABI, 1 and 4 samples, several
    views, restricted rectangles, every target drawn twice -- kernels and host code under AddressSanitizer
    (workgroup-local arrays and the heap) and UndefinedBehaviorSanitizer, with the default settings and with the
    device knobs.  Any report fails the run."""
    build_emulation_library()
    import build_emu
    exe = build_emu.build_sanitized()
    env = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    runs = [(k, subprocess.Popen([exe], env=dict(env, **k), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            for k in ({}, dict(SR_TILE_LOG2="6"), dict(SR_REF_CAPACITY="16", SR_SCRATCH_MIB="1"))]
    for k, proc in runs:
        out, _ = proc.communicate(timeout=1200)
        assert proc.returncode == 0 and "ok: 32 draws" in out and "runtime error" not in out and "AddressSanitizer" not in out, f"knobs {k}:\n{out[-3000:]}"


def test_differential_fuzzing_against_the_oracle():
    """tests/emu/fuzz_parity.py, a short run of each kind: random soups, dust, slivers and strips under
    random matrices, cameras (some inside the cloud), frame sizes, sample counts, rectangles and view
    counts.  Visibility is bit-exact, the clipped and given-up counters match and colour meets the bar on every
    scene, with trilinear and with 16x anisotropic samplers."""
    build_emulation_library()
    script = os.path.join(ROOT, "tests", "emu", "fuzz_parity.py")
    # the last two: every frame drawn twice (the second draw takes the lean and direct paths); batches of 9 to 14 views whose vertex and
    # set-up workgroups loop over the views, drawn twice (lean repeats store no depth plane)
    many = dict(SR_FUZZ_MANY_VIEWS="1", SR_FILL_BLOCKS="1", SR_TEST_DRAWS="2")
    kinds = ((dict(SR_FUZZ_ANISOTROPIC="0"), 0, 24), (dict(SR_FUZZ_ANISOTROPIC="0"), 100, 24), (dict(SR_FUZZ_ANISOTROPIC="0"), 2300, 24), (dict(SR_FUZZ_ANISOTROPIC="1"), 300, 24),
             (dict(SR_FUZZ_ANISOTROPIC="0"), 3400, 24), (dict(SR_FUZZ_ANISOTROPIC="1"), 3500, 24), (dict(SR_TEST_DRAWS="2", SR_STAGED_MIN_BLOCKS="1", SR_CULL_MIN_CHUNKS="1"), 5000, 24),
             (many, 7000, 10))
    runs = [(env, first, subprocess.Popen([sys.executable, script, str(first), str(count)], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
            for env, first, count in kinds]
    for env, first, proc in runs:
        out, _ = proc.communicate(timeout=1500)
        assert proc.returncode == 0 and "ok: seeds" in out, f"{env}, first seed {first}:\n{out[-3000:]}"


def test_benchmark_and_smoke_scripts_run_end_to_end():
    """tests/emu/rehearse.py: bench.py (three small workloads in one process, then its two multi-rank modes as two ranks over gloo) and __graft_entry__.smoke() against the CPU build
    of the kernels -- the scripts the round-end driver runs on the GPU box still produce their JSON line and their
    oracle comparison with the current library."""
    build_emulation_library()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "rehearse.py")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.count("bench ok:") == 5 and r.stdout.strip().endswith("smoke ok"), r.stdout[-3000:]


def test_file_readers_survive_mutated_files():
    """tests/emu/fuzz_files.py and fuzz_png.py, a short run each: the PNG and glTF readers under
    AddressSanitizer / UndefinedBehaviorSanitizer on mutated files."""
    for script in ("fuzz_files.py", "fuzz_png.py"):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", script), "3", "6"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0 and r.stdout.strip().splitlines()[-1].startswith("ok "), r.stdout[-3000:]


def test_out_of_range_access_is_reported(tmp_path):
    """The fence works: a kernel launched with a block shape its indexing does not expect is caught."""
    src = os.path.join(ROOT, "tests", "emu", "selftest_fault.cpp")
    exe = str(tmp_path / "selftest_fault")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "emu"), src,
                           os.path.join(ROOT, "tests", "emu", "hipemu.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0, r.stdout
    assert "memory access fault" in r.stdout and "kernel copy_kernel" in r.stdout and "past its end" in r.stdout, r.stdout
    assert "not reached by every live lane" in r.stdout and "kernel divergent_kernel" in r.stdout, r.stdout
