#!/usr/bin/env python3
"""Builds a variant of the GPU library with extra compiler flags for an A/B run:

    python tools/exp_build.py NAME [-DSR_SETUP_TRIS=4 ...]   ->  vulkan-renderer_amd/lib/exp/NAME.so

bench.py --lib PATH times a variant.  The variants are never loaded by anything else.

The gate: before anything is compiled for the GPU, the same sources are built for the CPU with the SAME -D flags
(tests/emu/build_emu.py, HIPEMU_EXTRA_FLAGS) and the gpu-marked parity suite is run on that build, where every device
allocation is fenced and uninitialised memory holds a pattern.  If that run fails, no GPU library is written and an
older one of the same name is deleted: a variant that indexes out of range (a table size a loop's trip count was not
written for, say) stops here instead of faulting a GPU.  There is no switch to skip the gate.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def gate(flags) -> bool:
    """The gpu-marked parity tests on the CPU build of the sources with `flags`, lanes in forward and in reversed order."""
    defines = [f for f in flags if f.startswith(("-D", "-U"))]
    env = dict(os.environ, SR_TEST_EMULATED="1", HIPEMU_EXTRA_FLAGS=" ".join(defines))
    build = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "emu", "build_emu.py")], env=env)
    if build.returncode != 0:
        print("exp_build: the CPU build of this variant does not compile", file=sys.stderr)
        return False
    for order in ("forward", "reverse"):
        cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), os.path.join(ROOT, "tests", "test_zz_no_alloc.py"),
               "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"]
        r = subprocess.run(cmd, cwd=ROOT, env=dict(env, HIPEMU_ORDER=order), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        tail = "\n".join(r.stdout.splitlines()[-25:])
        if r.returncode != 0 or " passed" not in tail or "failed" in tail:
            print(f"exp_build: the emulated parity suite FAILED for {defines or 'the default flags'} (lane order {order}); no GPU library written\n{tail}", file=sys.stderr)
            return False
    return True


def main():
    if len(sys.argv) < 2 or sys.argv[1].startswith("-"):
        raise SystemExit(__doc__)
    name, flags = sys.argv[1], sys.argv[2:]
    out = os.path.join(g.PKG, "lib", "exp", name + ".so")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    if not gate(flags):
        if os.path.exists(out):
            os.remove(out)
        raise SystemExit(1)
    srcs = [os.path.join(g.CSRC, f) for f in ("sr_kernels.hip", "sr_api.hip", "sr_assets.cpp", "sr_jpeg.cpp")]
    subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")] + g.HIPCC_FLAGS + flags + srcs + ["-lz", "-o", out])
    print(out)


if __name__ == "__main__":
    main()
