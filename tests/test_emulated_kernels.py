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

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_gpu_suite_emulated(order, select=None, timeout=900, what="tests", **extra_env):
    env = dict(os.environ, SR_TEST_EMULATED="1", HIPEMU_ORDER=order, **extra_env)
    cmd = [sys.executable, "-m", "pytest", os.path.join(ROOT, what), "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider"]
    if select:
        cmd += ["-k", select]
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    tail = "\n".join(r.stdout.splitlines()[-40:])
    assert r.returncode == 0, f"emulated run ({order}) failed:\n{tail}"
    assert " passed" in tail and "failed" not in tail, tail
    return tail


def test_gpu_parity_suite_on_emulated_kernels():
    """Whole gpu-marked suite, lanes scheduled in ascending order."""
    tail = run_gpu_suite_emulated("forward")
    # the two tests that need a real GPU process are the only ones allowed to be skipped
    assert "2 skipped" in tail, tail


@pytest.mark.parametrize("order", ["reverse", "shuffle"])
def test_results_do_not_depend_on_lane_order(order):
    """Same results when the lanes of a workgroup run in another order between synchronisation
    points: a missing barrier or an order-dependent reservation shows up as a parity failure."""
    run_gpu_suite_emulated(order, select="soup or huge or golden or split or multi_view or errors or trailing")


VARIANTS = ["fused_geometry", "dirty_blocks,exact_shading", "sparse_tail_force,dirty_blocks", "sparse_tail,dirty_blocks,exact_shading"]


@pytest.mark.parametrize("variant", VARIANTS)
def test_pipeline_variants_give_the_same_frames(variant):
    """The switchable pipeline variants (environment variable SR_VARIANT, see sr_api.hip) must pass the
    same parity suite as the default pipeline."""
    run_gpu_suite_emulated("forward", SR_VARIANT=variant)


def test_pipeline_variants_do_not_depend_on_lane_order():
    run_gpu_suite_emulated("reverse", select="reused or soup or golden or zz_full or huge", SR_VARIANT="sparse_tail_force,dirty_blocks,exact_shading")
    run_gpu_suite_emulated("shuffle", select="reused or soup or golden or zz_full or huge", SR_VARIANT="dirty_blocks")


def test_full_size_frame_on_emulated_kernels():
    """The 10 M-triangle 3840x2160 frame of the headline benchmark, executed on the CPU (about a
    minute), with every result of the hardware's approximate reciprocal / rsqrt / exp2 / log2 moved
    by up to one ulp: windows against the oracle plus the whole-frame properties."""
    run_gpu_suite_emulated("forward", what=os.path.join("tests", "test_gpu_zz_full_size.py"), timeout=1500,
                           SR_EMULATE_FULL_SIZE="1", HIPEMU_FUZZ_ULP="1")


def test_out_of_range_access_is_reported(tmp_path):
    """The fence works: a kernel launched with a block shape its indexing does not expect is caught."""
    src = os.path.join(ROOT, "tests", "emu", "selftest_fault.cpp")
    exe = str(tmp_path / "selftest_fault")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "tests", "emu"), src,
                           os.path.join(ROOT, "tests", "emu", "hipemu.cpp"), "-o", exe])
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=60)
    assert r.returncode == 0, r.stdout
    assert "memory access fault" in r.stdout and "kernel copy_kernel" in r.stdout and "past its end" in r.stdout, r.stdout
