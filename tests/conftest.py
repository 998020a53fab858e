import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "vulkan-renderer_amd"), os.path.dirname(os.path.abspath(__file__)), os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as o
    o.build()
    # the "anisotropic" pipeline variant is checked against the oracle's anisotropic footprint
    o.set_anisotropic("anisotropic" in os.environ.get("SR_VARIANT", ""))
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
    return softraster


@pytest.fixture(scope="session")
def device(sr):
    """One device handle for the whole GPU session.  No fallback: if the HIP library or the GPU is
    missing this raises, and the gpu-marked tests fail loudly."""
    dev = sr.Device(0)
    yield dev
    dev.close()
