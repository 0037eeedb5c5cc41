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
