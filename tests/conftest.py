import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vulkan-renderer_amd"), os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


# Within test_gpu_parity.py the tests of the basic draw paths are run before those of the grow-and-repeat,
# hash-overflow, pass-splitting, target-reuse and file-loading paths: the suite is run with -x, and a failure in
# one of the special paths should not hide the results of the basic ones.
RUN_AFTER_THE_BASIC_PARITY_TESTS = ("test_target_is_reused_across_different_draws", "test_mesh_and_textures_destroyed_before_the_read_back", "test_lean_draws_and_their_fallback",
                                    "test_model_and_textures_from_files", "test_tile_reference_list_grows_and_the_draw_is_repeated",
                                    "test_more_distinct_tiles_than_a_binning_block_can_hash", "test_view_batch_split_into_passes", "test_near_plane_clipper",
                                    "test_fragments_exactly_on_the_far_plane")


def pytest_collection_modifyitems(config, items):
    slots = [i for i, it in enumerate(items) if it.fspath.basename == "test_gpu_parity.py"]
    ordered = sorted((items[i] for i in slots), key=lambda it: it.name.split("[")[0] in RUN_AFTER_THE_BASIC_PARITY_TESTS)   # stable
    for i, it in zip(slots, ordered):
        items[i] = it


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    o.set_anisotropic(False)   # the session's default sampler is isotropic (see `sr` below); the `anisotropic` fixture switches both
    o.set_near_clip(True)      # the library clips at the near plane, as the reference's pipeline does
    return o


EMULATED = os.environ.get("SR_TEST_EMULATED") == "1"


@pytest.fixture(scope="session")
def sr():
    import softraster
    if EMULATED:
        # Set only by tests/test_emulated_kernels.py for its child process: the kernel sources compiled
        # for the host (tests/emu) stand in for the GPU library, so that the gpu-marked tests double as
        # a bounds and synchronisation check that needs no GPU.  Never set on a GPU box.
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))
        import build_emu
        softraster.LIB_PATH = build_emu.build()
        print(f"\n*** SR_TEST_EMULATED=1: kernels run on the CPU through {softraster.LIB_PATH} ***")
    # Textures made without an explicit sampler are trilinear (maxAnisotropy 1) in this session: the committed fixtures
    # and the headline configuration are.  Tests that take the `anisotropic` fixture run a second time with the
    # reference's own sampler (maxAnisotropy 16) on both sides.
    softraster.DEFAULT_MAX_ANISOTROPY = 1.0
    return softraster


@pytest.fixture(params=[False, True], ids=["trilinear", "aniso16"])
def anisotropic(request, sr, oracle):
    """Runs the test once with trilinear samplers and once with the reference's maxAnisotropy = 16
    (src/texture.cpp:255-256) in the library and in the oracle."""
    sr.DEFAULT_MAX_ANISOTROPY = 16.0 if request.param else 1.0
    oracle.set_anisotropic(request.param)
    yield request.param
    sr.DEFAULT_MAX_ANISOTROPY = 1.0
    oracle.set_anisotropic(False)


@pytest.fixture(scope="session")
def device(sr):
    """One device handle for the whole GPU session.  No fallback: if the HIP library or the GPU is
    missing this raises, and the gpu-marked tests fail loudly."""
    dev = sr.Device(0)
    yield dev
    dev.close()
