import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    # torch first: its bundled HIP runtime must be the one the process initialises (tests that hand torch device buffers to
    # the library found "No HIP GPUs are available" when the library's own runtime had been loaded before torch's)
    try:
        import torch

        if torch.cuda.is_available():
            torch.cuda.init()
    except Exception:
        pass
    import __graft_entry__ as g

    return g.load_package()


@pytest.fixture(scope="session")
def co():
    """C oracle (test infrastructure)."""
    from oracle import coracle

    coracle.lib()
    return coracle


@pytest.fixture(scope="session")
def o():
    """Python big-int oracle (test infrastructure)."""
    from oracle import bls12_381

    return bls12_381


@pytest.fixture(scope="session")
def ctx(pkg):
    c = pkg.Context([0])
    yield c
    c.close()


@pytest.fixture(scope="session")
def tctx(pkg):
    """Context on the TEST build of the library (libarkblst_amd_test.so: the same sources with -DMI_TEST_HOOKS)."""
    c = pkg.Context([0], test_hooks=True)
    yield c
    c.close()
