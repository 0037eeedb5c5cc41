import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "expects_fallback: the test provokes the persistent recurrent kernel's give-up path on purpose")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def real_ft_path():
    return os.path.join(GOLDEN, "final.feature_transform")


@pytest.fixture(scope="session")
def real_priors():
    return np.load(os.path.join(GOLDEN, "log_ap_Kaldi1909.npy"))


@pytest.fixture(scope="session")
def engine():
    """The HIP engine on device 0 -- GPU tests only; fails loudly if the .so or the GPU is missing."""
    from nnacousticmodeling_amd import engine as eng
    return eng.Engine(0)


@pytest.fixture(autouse=True)
def _no_silent_recurrent_fallback(request):
    """GPU tests: a persistent recurrent launch that gives up on a wait repeats the pass on the per-step kernels and still returns
    the right posteriors -- which would hide a broken arrival table behind a 2 s stall.  Every GPU test therefore also asserts that
    no such fallback happened (the one test that provokes it on purpose is marked `expects_fallback`)."""
    if request.node.get_closest_marker("gpu") is None:
        yield
        return
    from nnacousticmodeling_amd import _lib
    lib = _lib.load()
    before = lib.am_lstm_fallback_count()
    yield
    after = lib.am_lstm_fallback_count()
    if request.node.get_closest_marker("expects_fallback") is None:
        assert after == before, "the persistent recurrent kernel timed out and fell back to the per-step kernels"
